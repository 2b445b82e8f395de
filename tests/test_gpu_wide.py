"""Distance networks wider than the fused kernels' 256 columns (MLPRegression is width-agnostic, network_macros_mod.py:96-135):
the unfused GEMM path of csrc/wide_kernels.hip against the oracle -- raw forward / vjp rows, distance_repulsion_nn on a batch, a
propagate with injected samples, cost and update; ReLU and tanh, 384 / 512 / 1024 wide, ragged widths, through the reference-shaped
facade class as well."""
import numpy as np
import pytest

from helpers import OWN, RTOL, assert_close
from oracle import omds_oracle as orc

pytestmark = pytest.mark.gpu


def _net(widths, act, seed, n=7, C=9):
    """torch's nn.Linear default initialisation (uniform +- 1 / sqrt(fan_in)), seeded; distances in cm like the Franka net."""
    rng = np.random.RandomState(seed)
    dims = [3 * (n + 3)] + list(widths) + [C]
    W, b = [], []
    for i in range(len(dims) - 1):
        lim = 1.0 / np.sqrt(dims[i])
        W.append(rng.uniform(-lim, lim, (dims[i + 1], dims[i])).astype(np.float32))
        b.append(rng.uniform(-lim, lim, dims[i + 1]).astype(np.float32))
    W[-1] *= 40.0
    b[-1] = (b[-1] * 40.0 + 30.0).astype(np.float32)
    return orc.Mlp(W, b, act)


def _engine(m, N, H, k, obs):
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
    from optimalmodulationds_amd.engine import Engine
    e = Engine(7, N, H, k, max_obs=max(64, obs.shape[0]))
    e.set_mlp(m.W, m.b, act=m.act)
    e.set_obstacles(obs)
    e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.5, 0.01, 0b111
    e.push_params()
    e.set_ds(scenes.FRANKA_QF)
    e.set_cost(scenes.franka_dh_params(), np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32))
    return e


@pytest.mark.parametrize("widths,act", [((384, 384, 384), "relu"), ((512, 512), "tanh"), ((1024, 300, 257, 64), "relu")])
def test_wide_network_against_the_oracle(widths, act):
    from optimalmodulationds_amd import scenes
    m = _net(widths, act, seed=len(widths) * 7 + widths[0])
    obs = scenes.shelf_scene()
    N, H, k, K = 96, 4, 5, 3
    e = _engine(m, N, H, k, obs)
    assert not e.screen_stats()["active"]
    rng = np.random.RandomState(2)
    # raw rows: forward, arg-min link, vjp (robot_sdf.py:153-158)
    x = np.concatenate([scenes.FRANKA_Q0 + 0.5 * rng.standard_normal((200, 7)), rng.uniform(-0.5, 1.0, (200, 3))], 1).astype(np.float32)
    y, g, mi = e.mlp_forward_vjp(x)
    oy, og, omi = orc.mlp_vjp_argmin(m, x)
    assert_close(y, oy, RTOL, "raw forward", floor=OWN)
    assert (mi == omi).mean() >= 0.995                                    # near-ties of two links aside
    same = mi == omi
    assert_close(g[same], og[same], 2e-5, "vjp gradient", floor=float(np.abs(og).max()))   # every row: the trainer's GEMMs sum in the oracle's order
    # distance_repulsion_nn on a batch (MPPI.py:227-282)
    q = (scenes.FRANKA_Q0 + 0.4 * rng.standard_normal((N, 7))).astype(np.float32)
    dist, grad, mind, idx = e.dist_grad(q, want_mindist=True, want_idx=True)
    od, ogr, omind, oidx = orc.distance_repulsion_nn(m, q, obs, k, [0, 1, 2])
    assert_close(mind, omind, RTOL, "pass-1 matrix", floor=OWN)
    agree = (idx == oidx).all(axis=1)
    assert agree.mean() >= 0.97                                           # exact near-ties between two obstacles aside
    okq = agree
    assert_close(dist[agree], od[agree], RTOL, "closest distance", floor=OWN)
    assert_close(grad[okq], ogr[okq], 2e-5, "blended gradient", floor=float(np.abs(ogr).max()))
    # a propagate with injected samples, then cost and update on the device's own rollouts
    mu = (scenes.FRANKA_Q0 + 0.2 * rng.standard_normal((N, K, 7))).astype(np.float32)
    sg = np.ones((N, K), np.float32)
    al = rng.standard_normal((N, K, 7)).astype(np.float32)
    e.set_policy_samples(mu, sg, al)
    e.propagate(q)
    r = e.get_rollouts()
    ref = orc.propagate(m, q, scenes.FRANKA_QF, obs, N=N, H=H, dt=0.5, k=k, ignored_links=[0, 1, 2], mu_tmp=mu, sigma_tmp=sg, alpha_tmp=al,
                        prm=orc.Params(dst_thr=0.01))
    assert_close(r["closest_dist_all"][:, 0][agree], ref.closest_dist_all[:, 0][agree], RTOL, "step-1 distance", floor=OWN)
    assert_close(r["qdot"][okq], ref.qdot[okq], 2e-4, "step-1 velocity", floor=OWN)
    on = np.abs(r["all_traj"] - ref.all_traj).max(axis=(1, 2)) <= 1e-2
    assert on.mean() >= 0.9                                               # free-running: a few rollouts branch at rounding-level ties
    cost = e.cost()
    oc, _ = orc.evaluate_costs(r["all_traj"], r["closest_dist_all"], scenes.FRANKA_QF, scenes.franka_dh_params(), e_qmin(), e_qmax())
    assert_close(cost, oc, RTOL, "cost")
    e.close()


def e_qmin():
    from optimalmodulationds_amd.cost import FRANKA_Q_MIN
    return np.array(FRANKA_Q_MIN, np.float32)


def e_qmax():
    from optimalmodulationds_amd.cost import FRANKA_Q_MAX
    return np.array(FRANKA_Q_MAX, np.float32)


def test_wide_network_through_the_facade_and_its_limits():
    """RobotSdfCollisionNet(layers=[512] * 3) like the reference would build it; skip concatenations and widths above 4096 stay
    loud errors."""
    import torch
    from optimalmodulationds_amd import MPPI, LinDS, RobotSdfCollisionNet, scenes, _lib
    from optimalmodulationds_amd.engine import Engine
    m = _net((512, 512, 512), "relu", seed=3)
    nn_model = RobotSdfCollisionNet(in_channels=10, out_channels=9, layers=[512] * 3, skips=[])
    nn_model.model.W, nn_model.model.b = m.W, m.b      # (a checkpoint's state dict lands in the same two lists, robot_sdf.py:39-41)
    q_0, q_f = torch.tensor(scenes.FRANKA_Q0), torch.tensor(scenes.FRANKA_QF)
    dh = torch.tensor(scenes.franka_dh_params())
    mppi = MPPI(q_0, q_f, dh, torch.tensor(scenes.shelf_scene()), 0.5, 4, 64, [LinDS(q_f)], dh[:, 2], nn_model, 5)
    mppi.dst_thr = 0.01
    mppi.Policy.sample_policy()
    all_traj, dist, _, _, _ = mppi.propagate()
    cost = mppi.get_cost()
    assert torch.isfinite(all_traj).all() and cost.shape == (64,)
    d, g = mppi.distance_repulsion_nn(q_0[None])
    od, og, _, _ = orc.distance_repulsion_nn(m, q_0.numpy()[None], scenes.shelf_scene(), 5, [0, 1, 2])
    assert abs(float(d[0]) - float(od[0])) <= 1e-5 * max(1.0, abs(float(od[0])))
    e = Engine(7, 16, 2, 1, max_obs=8)
    with pytest.raises(_lib.OmdsError, match="4096"):
        big = _net((5000, 64), "relu", seed=1)
        e.set_mlp(big.W, big.b)
    with pytest.raises(_lib.OmdsError, match="skip"):
        rng = np.random.RandomState(0)
        W = [rng.standard_normal((300, 30)).astype(np.float32), rng.standard_normal((64, 330)).astype(np.float32), rng.standard_normal((9, 64)).astype(np.float32)]
        b = [np.zeros(300, np.float32), np.zeros(64, np.float32), np.zeros(9, np.float32)]
        e.set_mlp(W, b, skip_after=(0,))
    e.close()


def test_rejected_network_leaves_the_wide_one_installed():
    """A narrow network the library rejects (first layer of the wrong width) on a context that holds a wide one: the call fails
    with a status, and the wide network keeps answering with the numbers it gave before (the rejected call used to clear the
    wide state first, leaving a context whose fused kernels had null weight pointers)."""
    from optimalmodulationds_amd import scenes, _lib
    from optimalmodulationds_amd.engine import Engine
    m = _net((384, 384), "relu", seed=5)
    e = Engine(7, 32, 2, 5, max_obs=512)
    e.set_mlp(m.W, m.b)
    e.set_obstacles(scenes.shelf_scene())
    q = (scenes.FRANKA_Q0 + 0.2 * np.random.RandomState(1).standard_normal((32, 7))).astype(np.float32)
    d0, g0 = e.dist_grad(q)[:2]
    rng = np.random.RandomState(0)
    W = [rng.standard_normal((64, 29)).astype(np.float32), rng.standard_normal((9, 64)).astype(np.float32)]   # 29 != 3 (n + 3)
    with pytest.raises(_lib.OmdsError):
        e.set_mlp(W, [np.zeros(64, np.float32), np.zeros(9, np.float32)])
    d1, g1 = e.dist_grad(q)[:2]
    assert np.array_equal(d0, d1) and np.array_equal(g0, g1)
    e.set_ds(scenes.FRANKA_QF)
    e.set_policy_samples(np.zeros((32, 0, 7), np.float32), None, None)
    e.propagate(q)
    assert np.isfinite(e.get_rollouts()["all_traj"]).all()
    e.close()

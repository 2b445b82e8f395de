"""GPU tests of the fused one-launch step for scenes with few obstacles (csrc/step_small.hip, k_step_small): against the
two-kernel step (k_pass1 + k_tail, OMDS_FLAG_TWO_KERNEL_STEP), against the oracle at sampled states, over the shapes its
workgroup geometry distinguishes (R rollouts per workgroup = min(32 / O, 4 / k)).  The reference-captured planar fixtures
(planar2_*, planar7_*) run through it in test_gpu_parity.py as well -- it is the default step for them."""
import numpy as np
import pytest

from helpers import RTOL, assert_close, weights_path
from oracle import omds_oracle as orc

pytestmark = pytest.mark.gpu

KEYS = ("all_traj", "closest_dist_all", "kernel_val_all", "dot_products", "kernel_activations", "qdot", "normal")


def _scene(O, seed=7):
    from optimalmodulationds_amd import scenes
    base = scenes.planar7_scene(4, seed)
    if O <= base.shape[0]:
        return base[:O].copy()
    rng = np.random.RandomState(seed + 1)
    extra = np.c_[rng.uniform(-7, 7, (O - base.shape[0], 2)), np.zeros(O - base.shape[0]), np.full(O - base.shape[0], 0.5)]
    return np.concatenate((base, extra.astype(np.float32)))


def _engine(N, H, O, k, flags=0):
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path("planar7"))
    obs = _scene(O)
    q0 = np.zeros(7, np.float32); q0[0] = np.pi / 2
    qf = np.zeros(7, np.float32); qf[0] = -np.pi / 2
    e = Engine(7, N, H, k, max_obs=64, flags=flags)
    e.set_mlp(m.W, m.b)
    e.set_obstacles(obs)
    e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.3, 0.25, 0
    e.push_params()
    e.set_ds(qf)
    return e, m, obs, q0, qf


def _policy(rng, q0, qf, K):
    s = (np.arange(K) + 0.5) / max(K, 1)
    mu_c = (q0 + s[:, None] * (qf - q0) + 0.3 * rng.standard_normal((K, 7))).astype(np.float32)
    return mu_c, np.full(K, 0.5, np.float32), rng.standard_normal((K, 7)).astype(np.float32)


@pytest.mark.parametrize("N,H,O,k", [(1024, 4, 8, 1), (1022, 3, 8, 2), (257, 3, 5, 1), (64, 3, 32, 1), (96, 3, 1, 1),
                                     (128, 3, 16, 2), (40, 3, 3, 3), (513, 2, 12, 4)])
def test_fused_small_step_matches_the_two_kernel_step(N, H, O, k):
    """Same rollouts from both steps, to fp32 rounding: the fused step's forward is k_pass1's 32-row arithmetic, the
    two-kernel step runs its pass 2 (and for small batches its pass 1) on 16-row tiles -- another order of the k sums --
    and the 4-row backward sums in yet another order than the MFMA tiles."""
    from optimalmodulationds_amd import _lib as L
    outs, names, dg = [], [], []
    for flags in (0, L.FLAG_TWO_KERNEL_STEP):
        e, m, obs, q0, qf = _engine(N, H, O, k, flags)
        rng = np.random.RandomState(3)
        K = 6
        mu_c, sg_c, al_c = _policy(rng, q0, qf, K)
        e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 0.75, K, seed=21)
        q_start = (q0 + 0.4 * rng.standard_normal((N, 7))).astype(np.float32)     # per-rollout starts: a spread of states
        e.prof_enable(1)
        e.propagate(q_start)
        names.append(e.prof_read_ex()[3])
        outs.append(e.get_rollouts())
        dg.append(e.dist_grad(q_start[: min(N, 256)], want_mindist=True, want_idx=True))
        e.close()
    assert names[0] == "k_step_small" and names[1] == "k_pass1", names
    a, b = outs
    assert_close(a["closest_dist_all"][:, 0], b["closest_dist_all"][:, 0], 2e-6, "step 1 distance")
    for key in KEYS:
        x, y = (a[key], b[key]) if key == "qdot" else (a[key][:, 0], b[key][:, 0])      # qdot is the velocity of step 1
        assert_close(x, y, 5e-5, "step 1 " + key)      # two fp32 evaluations, each within 2e-5 of the oracle
        assert_close(a[key], b[key], 1e-2, "free-running " + key)    # later steps compound the rounding through sigmoids and ReLU masks (the oracle-vs-reference free-running bar)
    assert_close(a["all_traj"][:, 1], b["all_traj"][:, 1], 2e-5, "first integrated state")
    # the batch entry point follows the step of its context: same selected obstacles (up to exact near-ties), same numbers to rounding
    assert (dg[0][3] == dg[1][3]).mean() >= 0.999
    assert_close(dg[0][2], dg[1][2], 2e-6, "pass-1 matrix")
    assert_close(dg[0][0], dg[1][0], 2e-6, "blended distance")
    assert_close(dg[0][1], dg[1][1], 2e-5, "blended gradient", floor=float(np.abs(dg[1][1]).max()))


def test_fused_small_step_against_the_oracle():
    """planar 7-DoF, 8 obstacles, k = 1 (BASELINE configs[1] shape, 1024 x 32): sampled (rollout, step) states of a
    free-running propagate re-derived by the oracle -- distance, gradient direction, kernel values, integrated velocity."""
    N, H, O, k, K = 1024, 32, 8, 1, 10
    e, m, obs, q0, qf = _engine(N, H, O, k)
    rng = np.random.RandomState(11)
    mu_c, sg_c, al_c = _policy(rng, q0, qf, K)
    e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 0.75, K, seed=5)
    mu, sg, al = e.get_policy_samples()
    e.propagate(q0)
    r = e.get_rollouts()
    assert all(np.isfinite(v).all() for v in r.values())
    S = 384
    tt, hh = rng.randint(0, N, S), rng.randint(0, H, S)
    q = r["all_traj"][tt, hh]
    d, g, _, idx = orc.distance_repulsion_nn(m, q, obs, k, [])
    st = orc.modulation_step(q, qf, d, g, mu[tt], sg[tt], al[tt], orc.Params(dst_thr=0.25))
    scale = max(1.0, float(np.abs(d).max()))
    assert np.abs(r["closest_dist_all"][tt, hh] - (d - np.float32(0.25))).max() <= 1e-5 * scale
    assert_close(r["normal"][tt, hh], st["ghat"], 2e-5, "normal")          # every sampled row
    assert_close(r["kernel_val_all"][tt, hh], st["phi"], RTOL, "rbf")
    nxt = hh + 1 < H
    vel = (r["all_traj"][tt[nxt], hh[nxt] + 1] - q[nxt]) / np.float32(0.3)
    from helpers import assert_velocity_plain
    prm = orc.Params(dst_thr=0.25)
    assert_velocity_plain(vel, q[nxt], qf, d[nxt], g[nxt], mu[tt][nxt], sg[tt][nxt], al[tt][nxt], prm, "fused small step 1024 x 32, sampled rows",
                          pad=4e-6 * max(1.0, float(np.abs(q).max())) / 0.3, family="planar7")
    e.close()


def test_small_step_declines_what_it_cannot_do():
    """tanh networks, skip networks, more than 32 obstacles and k > 4 stay on the two-kernel step (loudly visible in the
    profiled kernel name), with the same API."""
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path("planar7"))
    q0 = np.zeros(7, np.float32); q0[0] = np.pi / 2
    for O, k, act in ((40, 1, "relu"), (8, 5, "relu"), (8, 1, "tanh")):
        e = Engine(7, 64, 2, k, max_obs=64)
        e.set_mlp(m.W, m.b, act=act)
        e.set_obstacles(_scene(O))
        e.set_ds(-q0)
        e.sample_policy(None, None, None, 0, 0, 0, 0, seed=1)
        e.prof_enable(1)
        e.propagate(q0)
        assert e.prof_read_ex()[3] == "k_pass1", (O, k, act)
        assert np.isfinite(e.get_rollouts()["all_traj"]).all()
        e.close()


@pytest.mark.parametrize("N,H,k", [(1, 2, 5), (40, 10, 5), (64, 4, 3), (83, 3, 5)])
def test_small_batches_take_pass_2s_forward_from_pass_1_and_keep_their_bits(N, H, k):
    """Up to 24 576 pairs the all-fp32 step of a ReLU network runs k_pass1 in its emitting mode (pass1_tile mode 6: every pair's
    pass-2 distance, arg-min link and ReLU masks beside Dmin) + k_tail_sel (backward only) instead of k_pass1 + k_tail (its own
    forward): the same numbers bit for bit as the context that keeps k_tail (OMDS_FLAG_TAIL_FORWARD)."""
    from optimalmodulationds_amd import scenes, _lib
    from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path("franka"))
    obs = scenes.shelf_scene()
    K = 6
    rng = np.random.RandomState(N)
    mu_c = (scenes.FRANKA_Q0 + 0.2 * rng.standard_normal((K, 7))).astype(np.float32)
    sg_c, al_c = np.ones(K, np.float32), rng.standard_normal((K, 7)).astype(np.float32)
    q = (scenes.FRANKA_Q0 + 0.3 * rng.standard_normal((N, 7))).astype(np.float32)
    outs = []
    for flags in (0, _lib.FLAG_TAIL_FORWARD):
        e = Engine(7, N, H, k, max_obs=296, flags=flags)
        e.set_mlp(m.W, m.b)
        e.set_obstacles(obs)
        e.set_screening(0)
        e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.5, 0.01, 0b111
        e.push_params()
        e.set_ds(scenes.FRANKA_QF)
        e.set_cost(scenes.franka_dh_params(), np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32))
        e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=3)
        e.propagate(q)
        r = e.get_rollouts()
        e.set_obstacles(obs[::3])               # another obstacle count on the same context: the per-pair buffers follow
        e.propagate(q)
        r2 = e.get_rollouts()
        outs.append((r, r2, e.cost()))
        e.close()
    for a, b in zip(outs[0][:2], outs[1][:2]):
        for key in a:
            assert np.array_equal(a[key], b[key]), key
    assert np.array_equal(outs[0][2], outs[1][2])
    assert np.isfinite(outs[0][0]["all_traj"]).all()


def test_dense_tail_sel_edge_cases():
    """the emitting pass 1 + k_tail_sel on rows where every pass-1 value ties (ALL links ignored: 1e6 everywhere, the
    selection falls back to the obstacle index) -- the same bits as the context that keeps k_tail, finite, and no counter touched (the
    dense mode has no window: its `viol` pointer is NULL); fewer obstacles than n_closest never reach a kernel (MPPI.py:245-247 would
    index past the sorted columns): omds_set_obstacles refuses them."""
    from optimalmodulationds_amd import scenes, _lib
    from optimalmodulationds_amd._lib import OmdsError
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path("franka"))
    obs = scenes.shelf_scene()
    q = (scenes.FRANKA_Q0 + 0.3 * np.random.RandomState(0).standard_normal((40, 7))).astype(np.float32)
    outs = []
    for flags in (0, _lib.FLAG_TAIL_FORWARD):
        e = Engine(7, 40, 3, 5, max_obs=296, flags=flags)
        e.set_mlp(m.W, m.b)
        e.set_obstacles(obs)
        e.set_screening(0)
        e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.5, 0.01, 0x1ff
        e.push_params()
        e.set_ds(scenes.FRANKA_QF)
        e.set_policy_samples(np.zeros((40, 0, 7), np.float32), np.zeros((40, 0), np.float32), np.zeros((40, 0, 7), np.float32))
        e.propagate(q)
        outs.append(e.get_rollouts())
        with pytest.raises(OmdsError, match="fewer obstacles than n_closest"):
            e.set_obstacles(obs[:3])
        e.close()
    assert np.isfinite(outs[0]["all_traj"]).all()
    for key in outs[0]:
        assert np.array_equal(outs[0][key], outs[1][key], equal_nan=True), key

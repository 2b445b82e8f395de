"""GPU tests of (a) the reference-shaped Python facade used like the reference's drivers use the
original classes, and (b) BASELINE.json's full sizes through size-independent properties."""
import numpy as np
import pytest
import torch

from helpers import RTOL, assert_close, load, weights_path
from oracle import omds_oracle as orc

pytestmark = pytest.mark.gpu


def _franka_mppi(N=64, H=6, k=5, obs=None):
    from optimalmodulationds_amd import MPPI, LinDS, RobotSdfCollisionNet, scenes
    nn_model = RobotSdfCollisionNet(in_channels=10, out_channels=9, layers=[256] * 4, skips=[])
    nn_model.load_weights(weights_path("franka"), {})
    nn_model.model_jit = nn_model
    nn_model.update_aot_lambda()
    q_0, q_f = torch.tensor(scenes.FRANKA_Q0), torch.tensor(scenes.FRANKA_QF)
    dh = torch.tensor(scenes.franka_dh_params())
    obs = torch.tensor(scenes.shelf_scene() if obs is None else obs)
    mppi = MPPI(q_0, q_f, dh, obs, 0.5, H, N, [LinDS(q_f), LinDS(q_0)], dh[:, 2], nn_model, k)
    mppi.Policy.sigma_c_nominal = 1
    mppi.Policy.alpha_s = 3
    mppi.Policy.policy_upd_rate = 0.5
    mppi.Policy.p = 2
    mppi.dst_thr = 0.01
    mppi.ker_thr = 0.1
    return mppi, nn_model


def test_planner_loop_like_the_reference_driver():
    """The call sequence of frankaPlanner.py:132-163 on the facade: shapes, attribute surface, kernels
    get added, and one iteration agrees with the oracle driven by the very same sampled tensors."""
    mppi, _ = _franka_mppi()
    N, H, n = 64, 6, 7
    dst_thr, thr_rbf_add, thr_dot_add = 0.03, 0.3, -0.9
    n_added = 0
    for it in range(6):
        mppi.Policy.sample_policy()
        all_traj, dist_all, kval, dots, acts = mppi.propagate()
        K = mppi.Policy.n_kernels
        assert all_traj.shape == (N, H, n) and dist_all.shape == (N, H) and kval.shape == (N, H, K)
        assert dots.shape == (N, H) and acts.shape == (N, H) and mppi.qdot.shape == (N, n)
        assert torch.equal(all_traj[:, 0, :], mppi.q_cur.expand(N, n))
        cost = mppi.get_cost()
        best_idx = torch.argmin(cost)
        assert torch.allclose(mppi.get_qdot('best'), mppi.qdot[best_idx], atol=1e-6)
        # --- oracle on the same samples ------------------------------------------------------------
        m = orc.Mlp.from_npz(weights_path("franka"))
        mu = mppi.Policy.mu_tmp[:, :K].numpy(); sg = mppi.Policy.sigma_tmp[:, :K].numpy(); al = mppi.Policy.alpha_tmp[:, :K].numpy()
        if K:
            assert np.array_equal(al[0], mppi.Policy.alpha_c[:K].numpy())          # policy.py:74
        o = orc.propagate(m, mppi.q_cur.numpy(), mppi.qf.numpy(), mppi.obs.numpy(), N=N, H=H, dt=0.5, k=5,
                          ignored_links=[0, 1, 2], mu_tmp=mu, sigma_tmp=sg, alpha_tmp=al, prm=orc.Params(dst_thr=0.01))
        assert_close(mppi.qdot.numpy(), o.qdot, 2e-4, "qdot vs oracle")
        assert_close(all_traj.numpy(), o.all_traj, 1e-2, "all_traj vs oracle (free running)")
        oc, _ = orc.evaluate_costs(all_traj.numpy(), dist_all.numpy(), mppi.qf.numpy(), mppi.dh_params.numpy(),
                                   mppi.Cost.q_min.numpy(), mppi.Cost.q_max.numpy())
        assert_close(cost.numpy(), oc, RTOL, "cost vs oracle")
        mu_c0, sg_c0, al_c0 = (x[:K].numpy().copy() for x in (mppi.Policy.mu_c, mppi.Policy.sigma_c, mppi.Policy.alpha_c))
        _, n_upd = mppi.shift_policy_means()
        omu, osg, oal, omask, _ = orc.shift_policy_means(cost.numpy(), kval.numpy(), acts.numpy(), mu_c0, sg_c0, al_c0,
                                                         mu, sg, al, 0.1, 0.1)
        assert n_upd == int(omask.sum())
        assert_close(mppi.Policy.alpha_c[:K].numpy(), oal, 2e-5, "alpha_c vs oracle")
        # --- kernel adding exactly as the driver does it --------------------------------------------
        cands = mppi.Policy.check_traj_for_kernels(all_traj, dist_all, dots, dst_thr - mppi.dst_thr, thr_rbf_add, thr_dot_add)
        if len(cands) > 0:
            norm, closest_idx = torch.norm(cands - mppi.q_cur, 2, -1).min(dim=0)
            idx_to_add = closest_idx if norm < 1e-1 else torch.randint(cands.shape[0], (1,))[0]
            cand = cands[idx_to_add]
            idx_i, idx_h = torch.where((all_traj == cand).all(dim=-1))
            mppi.Policy.add_kernel(cand, dist_all[idx_i[0], idx_h[0]], mppi.norm_basis[idx_i[0], idx_h[0]].squeeze())
            n_added += 1
        mppi.q_cur = mppi.q_cur + mppi.get_qdot('best') * 0.05
    assert n_added >= 1 and mppi.Policy.n_kernels == n_added
    assert mppi.norm_basis[3, 2].shape == (n, n)
    nb = mppi.norm_basis.tensor()
    assert nb.shape == (N, H, n, n)
    assert torch.equal(nb[3, 2], mppi.norm_basis[3, 2]) and torch.equal(nb[..., 0], mppi.normal_dirs)
    eye = torch.eye(n).expand(N, H, n, n)
    assert torch.allclose(nb.transpose(-1, -2) @ nb, eye, atol=2e-4)               # orthonormal, column 0 = normal
    # update_obstacles + update_kernel_normal_bases (frankaPlanner.py:125-130)
    obs2 = mppi.obs.clone(); obs2[:, 2] += 0.02
    mppi.update_obstacles(obs2)
    mppi.update_kernel_normal_bases()
    Kn = mppi.Policy.n_kernels
    B = mppi.Policy.kernel_obstacle_bases[:Kn]
    assert torch.allclose(B.transpose(-1, -2) @ B, torch.eye(n).expand(Kn, n, n), atol=2e-4)
    d, g = mppi.distance_repulsion_nn(mppi.Policy.mu_c[:Kn])
    od, og, _, _ = orc.distance_repulsion_nn(m, mppi.Policy.mu_c[:Kn].numpy(), obs2.numpy(), 5, [0, 1, 2])
    assert_close(d.numpy(), od, RTOL, "distance at kernel centres")
    # DS switching (frankaPlanner.py:118-122)
    mppi.switch_DS_idx(1)
    assert torch.equal(mppi.qf, mppi.DS_ARRAY[1].q_goal)
    mppi.Policy.reset_policy()
    mppi.Policy.sample_policy()
    mppi.propagate()
    assert float(mppi.get_cost().min()) >= 0


def test_integrator_shape_N1_H2():
    """frankaIntegratorSwitching.py:99-117: N=1, H=2, alpha_s = 0, policy installed from a dict."""
    from optimalmodulationds_amd import MPPI, LinDS, RobotSdfCollisionNet, scenes
    fx = load("franka_integrator_N1")
    nn_model = RobotSdfCollisionNet(10, 9, [], [256] * 4)
    nn_model.load_weights(weights_path("franka"), {})
    q0, qf = torch.tensor(fx["q0"]), torch.tensor(fx["qf"])
    dh = torch.tensor(fx["dh_params"])
    step = MPPI(q0, qf, dh, torch.tensor(fx["obs"]), 0.01, 2, 1, [LinDS(qf), LinDS(q0)], dh[:, 2], nn_model, 5)
    step.dst_thr = 0.03
    step.Policy.alpha_s *= 0
    K = int(fx["K"])
    step.Policy.update_with_data({"n_kernels": K, "mu_c": fx["it0_mu_c"], "alpha_c": fx["it0_alpha_c"],
                                  "sigma_c": fx["it0_sigma_c"], "norm_basis": np.zeros((K, 7, 7), np.float32)})
    step.Policy.sample_policy()            # alpha_s = 0 -> the samples equal the means
    step.q_cur = torch.tensor(fx["it0_q_cur"])
    step.propagate()
    assert step.qdot.shape == (1, 7)
    assert_close(step.qdot.numpy(), fx["it0_qdot"], 2e-4, "integrator qdot vs reference")
    q_new = torch.clamp(step.q_cur + step.qdot[0, :] * 0.01, step.Cost.q_min, step.Cost.q_max)
    assert q_new.shape == (7,)


def test_robot_sdf_facade_matches_reference_vectors():
    from optimalmodulationds_amd import RobotSdfCollisionNet
    fx = load("mlp_franka")
    nn_model = RobotSdfCollisionNet(10, 9, [], [256] * 4)
    nn_model.load_weights(weights_path("franka"), {})
    y = nn_model.model_jit.forward(torch.tensor(fx["x"]))
    assert_close(y.numpy(), fx["y"], RTOL, "forward")
    dists, grads, min_idx = nn_model.dist_grad_closest_aot(torch.tensor(fx["x"]))
    assert (min_idx.numpy() == fx["min_idx"]).all() and grads.shape == fx["grad"].shape


# ---------------------------------------------------------------------------------------------------------
# BASELINE sizes: Franka shelf, 1024 rollouts x 32 horizon (and 4096 x 4), size-independent properties
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big():
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path("franka"))
    N, H, k, K = 1024, 32, 5, 10
    obs = scenes.shelf_scene()
    eng = Engine(7, N, H, k, max_obs=512)
    eng.set_mlp(m.W, m.b)
    eng.set_obstacles(obs)
    eng.params.dt = 0.5; eng.params.dst_thr = 0.01; eng.params.ignored_links = 0b111
    eng.push_params()
    eng.set_ds(scenes.FRANKA_QF)
    from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
    eng.set_cost(scenes.franka_dh_params(), FRANKA_Q_MIN, FRANKA_Q_MAX)
    rng = np.random.RandomState(5)
    s = (np.arange(K) + 0.5) / K
    mu_c = (scenes.FRANKA_Q0 + s[:, None] * (scenes.FRANKA_QF - scenes.FRANKA_Q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
    sg_c = np.ones(K, np.float32)
    al_c = rng.standard_normal((K, 7)).astype(np.float32)
    eng.sample_policy(mu_c, sg_c, al_c, 0, 0, 3.0, K, seed=99)
    eng.propagate(scenes.FRANKA_Q0)
    r = eng.get_rollouts()
    yield dict(eng=eng, m=m, obs=obs, r=r, N=N, H=H, k=k, K=K, means=(mu_c, sg_c, al_c))
    eng.close()


def test_fullsize_properties(big):
    r, N, H = big["r"], big["N"], big["H"]
    for key in ("all_traj", "closest_dist_all", "qdot", "kernel_val_all"):
        assert np.isfinite(r[key]).all(), key
    g = r["normal"]
    assert np.abs(np.linalg.norm(g, axis=2) - 1).max() < 1e-5                       # unit normals
    assert ((r["kernel_val_all"] >= 0) & (r["kernel_val_all"] <= 1)).all()          # RBF values
    assert ((r["kernel_activations"] >= 0) & (r["kernel_activations"] <= 1 + 1e-6)).all()
    assert (np.abs(r["dot_products"]) <= 1 + 1e-5).all()
    # |u| <= 1 outside collision (normalised or <= 0.5), Euler consistency of the stored trajectory
    step = (r["all_traj"][:, 1:] - r["all_traj"][:, :-1]) / 0.5
    assert np.linalg.norm(step, axis=2).max() <= 1 + 1e-4
    assert_close(r["all_traj"][:, 1], r["all_traj"][:, 0] + 0.5 * r["qdot"], 1e-6, "first Euler step")
    # rollout 0 carries the mean policy; all rollouts start at q_cur
    assert np.abs(r["all_traj"][:, 0] - r["all_traj"][0, 0]).max() == 0


def test_fullsize_determinism_and_rollout_independence(big):
    """Two runs are bit-identical, and a rollout's result does not depend on its neighbours (rollouts
    are independent): re-running a 64-rollout subset through a small context reproduces it."""
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.engine import Engine
    eng, r = big["eng"], big["r"]
    eng.propagate(scenes.FRANKA_Q0)
    r2 = eng.get_rollouts()
    for key in r:
        assert np.array_equal(r[key], r2[key], equal_nan=True), f"{key} not deterministic"
    mu, sg, al = eng.get_policy_samples()
    sel = np.arange(0, big["N"], 16)
    small = Engine(7, len(sel), big["H"], big["k"], max_obs=512)
    small.set_mlp(big["m"].W, big["m"].b)
    small.set_obstacles(big["obs"])
    small.params.dt = 0.5; small.params.dst_thr = 0.01; small.params.ignored_links = 0b111
    small.push_params()
    small.set_ds(scenes.FRANKA_QF)
    small.set_policy_samples(mu[sel], sg[sel], al[sel])
    small.propagate(scenes.FRANKA_Q0)
    rs = small.get_rollouts()
    assert_close(rs["qdot"], r["qdot"][sel], 1e-6, "subset qdot")
    assert_close(rs["closest_dist_all"][:, 0], r["closest_dist_all"][sel, 0], 1e-6, "subset distance")
    small.close()


def test_fullsize_sampled_rows_against_oracle(big):
    """Oracle on a 48-rollout sample of the full-size run, teacher-forced at three horizon steps."""
    from optimalmodulationds_amd import scenes
    eng, r, m, obs = big["eng"], big["r"], big["m"], big["obs"]
    mu, sg, al = eng.get_policy_samples()
    sel = np.linspace(0, big["N"] - 1, 48).astype(int)
    for h in (0, 13, 30):
        q = r["all_traj"][sel, h]
        d, g, mind, idx = orc.distance_repulsion_nn(m, q, obs, big["k"], [0, 1, 2])
        assert_close(r["closest_dist_all"][sel, h], d - np.float32(0.01), RTOL, f"distance h={h}")
        st = orc.modulation_step(q, scenes.FRANKA_QF, d, g, mu[sel], sg[sel], al[sel], orc.Params(dst_thr=0.01))
        ok = orc.rollout_relu_margin(m, q, obs, idx) >= 5e-6
        assert ok.mean() > 0.5
        assert_close(r["normal"][sel, h][ok], st["ghat"][ok], 5e-5, f"normal h={h}")
        assert_close((r["all_traj"][sel, h + 1] - q)[ok] / 0.5, st["u"][ok], 5e-4, f"velocity h={h}")


def test_fullsize_obstacle_permutation_invariance(big):
    """Shuffling the obstacle list permutes the pass-1 matrix columns and leaves distances unchanged."""
    from optimalmodulationds_amd import scenes
    eng, r, obs = big["eng"], big["r"], big["obs"]
    q = r["all_traj"][:256, 7]
    d1, g1, m1, i1 = eng.dist_grad(q, want_mindist=True, want_idx=True)
    perm = np.random.RandomState(0).permutation(obs.shape[0])
    eng.set_obstacles(obs[perm])
    d2, g2, m2, i2 = eng.dist_grad(q, want_mindist=True, want_idx=True)
    eng.set_obstacles(obs)
    assert np.array_equal(m2, m1[:, perm])                                          # same rows, same arithmetic
    assert np.array_equal(np.sort(perm[i2], axis=1), np.sort(i1, axis=1)) or \
        np.allclose(np.take_along_axis(m1, i1.astype(np.int64), 1), np.take_along_axis(m2, i2.astype(np.int64), 1))
    assert_close(d2, d1, 1e-6, "distance under obstacle permutation")


def test_fullsize_update_sums(big):
    eng, K = big["eng"], big["K"]
    mu_c, sg_c, al_c = big["means"]
    eng.cost()
    mu, sg, al, mask, w = eng.weighted_update(0.1, 0.1, mu_c, sg_c, al_c, want_weights=True)
    assert abs(float(w.sum(dtype=np.float64)) - 1) < 1e-5 and (w >= 0).all()
    assert np.array_equal(mu[~mask], mu_c[~mask]) and np.array_equal(al[~mask], al_c[~mask])
    assert_close(mu, mu_c, 1e-6, "mu_c unchanged when mu_s = 0")
    cs = eng.cost_sum()
    assert cs[1] == big["N"]


def _mk(N, H, k, obs, m):
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.engine import Engine
    eng = Engine(7, N, H, k, max_obs=max(8, obs.shape[0]))
    eng.set_mlp(m.W, m.b, act=m.act)
    eng.set_obstacles(obs)
    eng.params.dt = 0.5; eng.params.dst_thr = 0.01; eng.params.ignored_links = 0b111
    eng.push_params()
    eng.set_ds(scenes.FRANKA_QF)
    return eng


@pytest.mark.parametrize("N,H,k,O", [(1000, 3, 5, 294), (77, 2, 7, 33), (8192, 2, 5, 294), (130, 2, 5, 5), (65, 3, 1, 1),
                                     # tiny batches: 16-row pass-1 / pass-2 tiles (integrator shape, k = 16 and k = 17 around the
                                     # 16-row limit, a single pair)
                                     (1, 2, 5, 294), (1, 1, 1, 1), (3, 2, 2, 2), (17, 2, 16, 40), (9, 2, 17, 40), (40, 3, 5, 294)])
def test_ragged_and_large_shapes(N, H, k, O):
    """Sizes that are not multiples of any tile (rows per pass-1 tile 64/32/16, rollouts per tail workgroup
    floor(32/k) or floor(16/k)), O == k, O == 1, and a large N: a 40-rollout sample must match the oracle step by step."""
    from optimalmodulationds_amd import scenes
    m = orc.Mlp.from_npz(weights_path("franka"))
    obs = scenes.shelf_scene()[np.linspace(0, 293, O).astype(int)]
    rng = np.random.RandomState(N)
    K = 3
    eng = _mk(N, H, k, obs, m)
    q0 = (scenes.FRANKA_Q0 + 0.4 * rng.standard_normal((N, 7))).astype(np.float32)
    mu = (scenes.FRANKA_Q0 + 0.3 * rng.standard_normal((N, K, 7))).astype(np.float32)
    sg = np.ones((N, K), np.float32)
    al = rng.standard_normal((N, K, 7)).astype(np.float32)
    eng.set_policy_samples(mu, sg, al)
    eng.propagate(q0)
    r = eng.get_rollouts()
    assert np.isfinite(r["all_traj"]).all() and np.isfinite(r["closest_dist_all"]).all()
    sel = np.unique(np.concatenate(([0, min(1, N - 1), max(N - 2, 0), N - 1], rng.choice(N, min(36, N), replace=False))))
    for h in range(H):
        q = r["all_traj"][sel, h]
        d, g, _, idx = orc.distance_repulsion_nn(m, q, obs, k, [0, 1, 2])
        assert_close(r["closest_dist_all"][sel, h], d - np.float32(0.01), RTOL, f"distance h={h}")
        st = orc.modulation_step(q, scenes.FRANKA_QF, d, g, mu[sel], sg[sel], al[sel], orc.Params(dst_thr=0.01))
        ok = orc.rollout_relu_margin(m, q, obs, idx) >= 5e-6
        assert_close(r["normal"][sel, h][ok], st["ghat"][ok], 5e-5, f"normal h={h}")
        if h + 1 < H:
            assert_close((r["all_traj"][sel, h + 1] - q)[ok] / 0.5, st["u"][ok], 5e-4, f"velocity h={h}")
    eng.close()


def test_example_drivers_run():
    """examples/: the reference's two driver loops (Franka planner, planar 2-DoF stand-alone) on the facade."""
    import importlib.util
    import os
    from helpers import ROOT
    mods = {}
    for name in ("franka_planner_loop", "standalone_planar2d"):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
        mods[name] = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mods[name])
    mppi = mods["franka_planner_loop"].main(iters=6, n_traj=128, horizon=8, moving=True, quiet=True)
    assert torch.isfinite(mppi.q_cur).all() and mppi.Policy.n_kernels <= 6
    mppi2, n_iter = mods["standalone_planar2d"].main(max_iter=60, quiet=True)
    d0 = float(torch.norm(torch.tensor([-3.14, 0.0]) - torch.tensor([3.14, 0.0])))
    assert float(torch.norm(mppi2.q_cur - torch.tensor([3.14, 0.0]))) < d0        # it moves towards the goal

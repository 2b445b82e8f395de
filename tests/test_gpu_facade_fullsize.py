"""GPU tests of (a) the reference-shaped Python facade used like the reference's drivers use the
original classes, and (b) BASELINE.json's full sizes through size-independent properties."""
import numpy as np
import pytest
import torch

from helpers import OWN, RTOL, assert_close, assert_velocity_plain, load, weights_path
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
    assert_close(d.numpy(), od, RTOL, "distance at kernel centres", floor=OWN)
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
    assert_close(rs["qdot"], r["qdot"][sel], 1e-6, "subset qdot", floor=OWN)
    assert_close(rs["closest_dist_all"][:, 0], r["closest_dist_all"][sel, 0], 1e-6, "subset distance", floor=OWN)
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
        assert_close(r["closest_dist_all"][sel, h], d - np.float32(0.01), RTOL, f"distance h={h}", floor=OWN)
        st = orc.modulation_step(q, scenes.FRANKA_QF, d, g, mu[sel], sg[sel], al[sel], orc.Params(dst_thr=0.01))
        assert_close(r["normal"][sel, h], st["ghat"], 2e-5, f"normal h={h}")          # every sampled row
        vel = (r["all_traj"][sel, h + 1] - q) / np.float32(0.5)
        assert_velocity_plain(vel, q, scenes.FRANKA_QF, d, g, mu[sel], sg[sel], al[sel], orc.Params(dst_thr=0.01), f"velocity h={h}",
                              pad=4e-6 * max(1.0, float(np.abs(q).max())) / 0.5)


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
    assert_close(d2, d1, 1e-6, "distance under obstacle permutation", floor=OWN)


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
    eng.set_mlp(m.W, m.b, act=m.act, skip_after=m.skip_after)
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
        assert_close(r["closest_dist_all"][sel, h], d - np.float32(0.01), RTOL, f"distance h={h}", floor=OWN)
        st = orc.modulation_step(q, scenes.FRANKA_QF, d, g, mu[sel], sg[sel], al[sel], orc.Params(dst_thr=0.01))
        assert_close(r["normal"][sel, h], st["ghat"], 2e-5, f"normal h={h}")
        if h + 1 < H:
            vel = (r["all_traj"][sel, h + 1] - q) / np.float32(0.5)
            assert_velocity_plain(vel, q, scenes.FRANKA_QF, d, g, mu[sel], sg[sel], al[sel], orc.Params(dst_thr=0.01), f"velocity h={h}",
                                  pad=4e-6 * max(1.0, float(np.abs(q).max())) / 0.5)
    eng.close()


def test_example_drivers_run():
    """examples/: the reference's driver loops (Franka planner, planar 2-DoF and planar 7-DoF stand-alone) on the facade."""
    import importlib.util
    import os
    from helpers import ROOT
    mods = {}
    for name in ("franka_planner_loop", "standalone_planar2d", "standalone_planar7d"):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
        mods[name] = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mods[name])
    mppi = mods["franka_planner_loop"].main(iters=6, n_traj=128, horizon=8, moving=True, quiet=True)
    assert torch.isfinite(mppi.q_cur).all() and mppi.Policy.n_kernels <= 6
    mppi2, n_iter = mods["standalone_planar2d"].main(max_iter=60, quiet=True)
    d0 = float(torch.norm(torch.tensor([-3.14, 0.0]) - torch.tensor([3.14, 0.0])))
    assert float(torch.norm(mppi2.q_cur - torch.tensor([3.14, 0.0]))) < d0        # it moves towards the goal
    # standalonePlanar7d.py:96-166 with its own defaults (20 rollouts x 10 steps, dt_sim = 0.02): 60 iterations move q[0] from pi/2 towards -pi/2
    mppi7, n7 = mods["standalone_planar7d"].main(max_iter=60, quiet=True)
    q_f7 = torch.zeros(7); q_f7[0] = -torch.pi / 2
    assert n7 == 60 and torch.isfinite(mppi7.q_cur).all() and float(torch.norm(mppi7.q_cur - q_f7)) < float(torch.pi) - 0.5
    # ... and at BASELINE configs[1]'s shape (1024 rollouts x 32 steps, 8 obstacles) for a few iterations
    mppi7b, _ = mods["standalone_planar7d"].main(max_iter=3, n_traj=1024, dt_h=32, n_extra_obs=4, quiet=True)
    assert mppi7b.all_traj.shape == (1024, 32, 7) and torch.isfinite(mppi7b.q_cur).all()


def test_planner_payload_round_trip_into_the_integrator():
    """frankaPlanner.py:166-179 -> (pickle) -> frankaIntegratorSwitching.py:102-117: the policy dictionary a planner on
    this package publishes is consumed by an N = 1, H = 2 integrator MPPI; its tick equals the oracle's step under the same
    policy, and the state dictionary has the reference's keys."""
    import pickle
    from optimalmodulationds_amd import MPPI, LinDS, RobotSdfCollisionNet, scenes
    from optimalmodulationds_amd.payloads import integrator_tick, kernel_fk, planner_payload
    mppi, nn_model = _franka_mppi(N=64, H=6)
    # a planner iteration with two kernels installed the way add_kernel does it
    all_kernel_fk = []
    for c in (0.3, 0.6):
        q_k = mppi.q0 + c * (mppi.qf - mppi.q0)
        d, g = mppi.distance_repulsion_nn(q_k[None])
        E = torch.linalg.qr(torch.cat((g.reshape(-1, 1), torch.eye(7)[:, 1:]), dim=1))[0]
        mppi.Policy.add_kernel(q_k, d[0], E)
        all_kernel_fk.append(kernel_fk(q_k, mppi.dh_params))
    mppi.Policy.sample_policy()
    mppi.propagate()
    cost = mppi.get_cost()
    mppi.shift_policy_means()
    data = pickle.loads(pickle.dumps(planner_payload(mppi, cost, all_kernel_fk)))       # the wire format is a pickle
    K = mppi.Policy.n_kernels
    assert set(data) == {'n_kernels', 'mu_c', 'alpha_c', 'sigma_c', 'norm_basis', 'kernel_fk', 'best_traj_fk'}
    assert data['n_kernels'] == K == 2 and data['mu_c'].shape == (K, 7) and data['norm_basis'].shape == (K, 7, 7)
    assert data['best_traj_fk'].shape == (1, 14, 3) and data['kernel_fk'][0].shape == (12, 3)
    best = int(torch.argmin(cost))
    from optimalmodulationds_amd.fk_num import numeric_fk_model
    assert torch.allclose(data['best_traj_fk'][0], numeric_fk_model(mppi.all_traj[best, -1], mppi.dh_params, 2)[0].reshape(-1, 3), atol=1e-6)
    # integrator side
    dh = mppi.dh_params
    step = MPPI(mppi.q0, mppi.qf, dh, mppi.obs, 0.01, 2, 1, [LinDS(mppi.qf), LinDS(mppi.q0)], dh[:, 2], nn_model, 5)
    step.dst_thr = 0.03
    step.Policy.alpha_s *= 0
    step.Policy.p = mppi.Policy.p
    q_before = step.q_cur.clone()
    state = integrator_tick(step, data, mppi.obs, 0.01)
    assert set(state) == {'q', 'dq', 'ds_idx'} and state['q'].shape == (7,) and state['dq'].shape == (7,) and state['ds_idx'] == 0
    assert step.Policy.n_kernels == K and torch.equal(step.Policy.mu_c[:K], data['mu_c'])
    assert torch.equal(step.Policy.kernel_obstacle_bases[:K], data['norm_basis'])
    m = orc.Mlp.from_npz(weights_path("franka"))
    o = orc.propagate(m, q_before.numpy(), mppi.qf.numpy(), mppi.obs.numpy(), N=1, H=2, dt=0.01, k=5, ignored_links=[0, 1, 2],
                      mu_tmp=data['mu_c'].numpy()[None], sigma_tmp=data['sigma_c'].numpy()[None], alpha_tmp=data['alpha_c'].numpy()[None],
                      prm=orc.Params(dst_thr=0.03))
    assert_close(state['dq'].numpy(), o.qdot[0], 2e-4, "integrator dq vs oracle")
    assert torch.allclose(state['q'], torch.clamp(q_before + state['dq'] * 0.01, step.Cost.q_min, step.Cost.q_max))


def test_cost_of_foreign_tensors_and_obstacle_growth():
    """Cost.evaluate_costs evaluates the tensors it is given (cost.py:13-22), not only the owner's rollouts; and
    update_obstacles accepts a set larger than anything seen so far (MPPI.py:347-350, frankaPlanner.py:126)."""
    from optimalmodulationds_amd import scenes
    small = scenes.shelf_scene()[:20]
    mppi, _ = _franka_mppi(N=64, H=6, obs=small)
    assert mppi._max_obs < 294
    mppi.Policy.sample_policy()
    all_traj, dist_all, *_ = mppi.propagate()
    c_own = mppi.Cost.evaluate_costs(all_traj, dist_all)
    oc, _ = orc.evaluate_costs(all_traj.numpy(), dist_all.numpy(), mppi.qf.numpy(), mppi.dh_params.numpy(),
                               mppi.Cost.q_min.numpy(), mppi.Cost.q_max.numpy())
    assert_close(c_own.numpy(), oc, RTOL, "own rollouts")
    # edited trajectories / a subset: evaluated as given
    traj2 = all_traj[5:40].clone()
    traj2[:, -1, :] += 0.05
    dist2 = dist_all[5:40].clone()
    dist2[::3, 2] = -0.01
    c2 = mppi.Cost.evaluate_costs(traj2, dist2)
    oc2, _ = orc.evaluate_costs(traj2.numpy(), dist2.numpy(), mppi.qf.numpy(), mppi.dh_params.numpy(),
                                mppi.Cost.q_min.numpy(), mppi.Cost.q_max.numpy())
    assert c2.shape == (35,)
    assert_close(c2.numpy(), oc2, RTOL, "foreign tensors")
    assert torch.equal(mppi.Cost.evaluate_costs(all_traj, dist_all), c_own)        # the device rollouts were not disturbed
    # a much larger obstacle set arrives mid-loop
    big_obs = torch.tensor(scenes.shelf_scene())
    mu, sg, al = (x.clone() for x in (mppi.Policy.mu_tmp, mppi.Policy.sigma_tmp, mppi.Policy.alpha_tmp))
    mppi.update_obstacles(big_obs)
    assert mppi._max_obs >= 294 and mppi.n_obs == 294
    all_traj, dist_all, *_ = mppi.propagate()
    m = orc.Mlp.from_npz(weights_path("franka"))
    K = mppi.Policy.n_kernels
    o = orc.propagate(m, mppi.q_cur.numpy(), mppi.qf.numpy(), big_obs.numpy(), N=64, H=6, dt=0.5, k=5, ignored_links=[0, 1, 2],
                      mu_tmp=mu[:, :K].numpy(), sigma_tmp=sg[:, :K].numpy(), alpha_tmp=al[:, :K].numpy(), prm=orc.Params(dst_thr=0.01))
    assert_close(mppi.qdot.numpy(), o.qdot, 2e-4, "qdot after the capacity grew")


def _basis_tol(ghat0):
    """QR of [g, e_2 .. e_n] is a Householder reflection built from g + sign(g_0) |g| e_1: its tangent columns move by
    ~ delta(g) / (1 + |g_0|/|g|) -- benign -- except through the sign choice at g_0 = 0; column 0 is overwritten by g."""
    return np.minimum(2e-2, 5e-5 * (1.0 + 1.0 / np.maximum(np.abs(ghat0), 1e-6)))


@pytest.mark.parametrize("name", ["franka_shelf_K6", "franka_sub40_K4", "planar7_K4", "planar2_c1_K3"])
def test_norm_basis_matches_the_reference_qr(name):
    """MPPI.norm_basis (MPPI.py:122-127): the facade's lazily completed basis vs the reference's own [N,H,n,n] tensor,
    every horizon step restarted from the reference's state (so that the normals are comparable entry by entry)."""
    from optimalmodulationds_amd.mppi import _qr_complete
    from test_gpu_parity import _engine
    fx = load(name)
    if "it0_norm_basis" not in fx:
        pytest.skip("fixture without the full basis")
    eng, m = _engine(fx, H=1)
    ref_nb, ref_traj = fx["it0_norm_basis"], fx["it0_all_traj"]
    K = int(fx["K"])
    eng.set_policy_samples(fx["it0_mu_tmp"][:, :K], fx["it0_sigma_tmp"][:, :K], fx["it0_alpha_tmp"][:, :K])
    worst = 0.0
    for h in range(int(fx["H"])):
        eng.propagate(np.ascontiguousarray(ref_traj[:, h, :]))
        normal = eng.get_rollouts(want=("normal",))["normal"][:, 0]            # [N, n]
        E = _qr_complete(normal).numpy()
        R = ref_nb[:, h]
        finite = np.isfinite(R).all(axis=(1, 2)) & np.isfinite(E).all(axis=(1, 2))
        assert finite.mean() > 0.9
        assert np.abs(E[finite][:, :, 0] - R[finite][:, :, 0]).max() <= 5e-5, "column 0 = the obstacle normal"
        tol = _basis_tol(R[finite][:, 0, 0])
        err = np.abs(E[finite] - R[finite]).max(axis=(1, 2))
        assert (err <= tol).all(), (h, float(err.max()), float(tol[np.argmax(err - tol)]))
        worst = max(worst, float(err.max()))
        eye = np.einsum('tij,til->tjl', E[finite], E[finite])
        assert np.abs(eye - np.eye(E.shape[-1])).max() < 2e-4
    eng.close()


def test_update_kernel_normal_bases_matches_the_reference():
    """MPPI.update_kernel_normal_bases (MPPI.py:284-304) after the obstacles moved: Policy.kernel_obstacle_bases vs the
    tensor the reference produced for the same kernel centres and obstacles (tests/golden/bases_franka.npz,
    tools/make_golden_bases.py)."""
    fx = load("bases_franka")
    mppi, _ = _franka_mppi(N=8, H=2)
    K = int(fx["K"])
    mppi.Policy.n_kernels = K
    mppi.Policy.mu_c[:K] = torch.tensor(fx["mu_c"])
    mppi.update_obstacles(torch.tensor(fx["obs"]))
    mppi.update_kernel_normal_bases()
    B = mppi.Policy.kernel_obstacle_bases[:K].numpy()
    d, g = mppi.distance_repulsion_nn(mppi.Policy.mu_c[:K])
    assert_close(d.numpy(), fx["distance"], RTOL, "distance at the kernel centres", floor=OWN)
    assert_close(g.numpy(), fx["nn_grad"], 1e-4, "gradient at the kernel centres", floor=float(np.abs(fx["nn_grad"]).max()))
    assert np.abs(B[:, :, 0] - fx["bases"][:, :, 0]).max() <= 5e-5
    err = np.abs(B - fx["bases"]).max(axis=(1, 2))
    assert (err <= _basis_tol(fx["bases"][:, 0, 0])).all(), err


def test_integrator_with_a_seds_nominal_ds():
    """frankaIntegrator.py:70-73 with the SEDS lines un-commented: DS_ARRAY = [SEDS(<.mat>, q_f)], N = 1, H = 2.  The SEDS object
    is built from the mixture arrays of the reference's seds_left10.mat (data in the fixture), derives its per-component
    quantities like the reference does, and the step reproduces the reference's captured velocity."""
    from optimalmodulationds_amd import MPPI, SEDS, RobotSdfCollisionNet
    fx, mat = load("franka_seds_integrator_N1"), load("seds_left10")
    nn_model = RobotSdfCollisionNet(10, 9, [], [256] * 4)
    nn_model.load_weights(weights_path("franka"), {})
    q0, qf = torch.tensor(fx["q0"]), torch.tensor(fx["qf"])
    ds = SEDS({k: mat[k] for k in ("Mu", "Sigma", "Priors", "xT")}, qf.unsqueeze(1))
    for got, want in zip(ds.device_params(), (fx["seds_mu_in"], fx["seds_b"], fx["seds_sigma_inv"], fx["seds_A"], fx["seds_prior"], fx["seds_den"])):
        # torch.inverse in float32 of covariances with condition numbers of 10^3 .. 10^4: the derived arrays differ between hosts
        # (4e-5 between the capture container and the test box) -- as they would for the reference itself
        assert_close(got, want, 3e-4, "derived SEDS quantities", floor=float(np.abs(want).max()))
    dh = torch.tensor(fx["dh_params"])
    step = MPPI(q0, qf, dh, torch.tensor(fx["obs"]), 0.01, 2, 1, [ds], dh[:, 2], nn_model, 5)
    step.dst_thr = 0.03
    step.Policy.alpha_s *= 0
    K = int(fx["K"])
    step.Policy.update_with_data({"n_kernels": K, "mu_c": fx["it0_mu_c"], "alpha_c": fx["it0_alpha_c"],
                                  "sigma_c": fx["it0_sigma_c"], "norm_basis": np.zeros((K, 7, 7), np.float32)})
    step.Policy.sample_policy()
    step.q_cur = torch.tensor(fx["it0_q_cur"])
    step.propagate()
    assert_close(step.qdot.numpy(), fx["it0_qdot"], 1e-3, "integrator qdot with SEDS vs reference (host-derived mixture quantities)")
    # the host convenience method against the reference's known answers (one state per reference call)
    ds0 = SEDS({k: mat[k] for k in ("Mu", "Sigma", "Priors", "xT")})
    y = ds0.get_velocity(torch.tensor(mat["x"])).numpy()
    ok = np.abs(y - mat["y"]).max(axis=1) <= 5e-5 * max(1.0, float(np.abs(mat["y"]).max()))
    assert ok.mean() > 0.97     # a couple of states far outside the demonstrations are decided by denormal exponentials


def test_lazy_rollout_tensors_fetch_rows_and_behave_like_tensors():
    """With ``lazy_rollouts = True`` ``propagate()`` returns LazyRollout tensors (lazy.py): indexing by rollout fetches only those rows
    (omds_get_rollout_rows) and equals the same index into the fully fetched tensor; every other use behaves like the torch tensor of
    the default (eager) form; a tensor of an earlier propagate that was never read refuses to hand out the next propagate's numbers;
    writing into one and pickling one materialise it.  The default hands out and binds plain torch tensors like the reference."""
    import pickle
    from optimalmodulationds_amd.lazy import LazyRollout
    mppi, _ = _franka_mppi()
    mppi.Policy.sample_policy()
    eager = mppi.propagate()                                               # the class default: the reference's plain tensors
    assert all(isinstance(x, torch.Tensor) for x in eager) and eager[0] is mppi.all_traj and isinstance(mppi.qdot, torch.Tensor)
    cand_dev = mppi.Policy.check_traj_for_kernels(eager[0], eager[1], eager[3], 0.3, 0.5, 0.5)      # device search on the unmodified tensors
    eager[1][0, 0] = eager[1][0, 0]                                        # an in-place write: the host path from here on, same answer
    assert not mppi._is_device_copy(eager[1])
    cand_host = mppi.Policy.check_traj_for_kernels(eager[0], eager[1], eager[3], 0.3, 0.5, 0.5)
    assert torch.equal(cand_dev, cand_host)
    kept_eager = eager[0].clone()
    mppi.Policy.sample_policy()
    mppi.propagate()
    assert torch.equal(eager[0], kept_eager)                               # a tensor kept across propagate() calls keeps its numbers
    mppi.lazy_rollouts = True
    N, H, n = 64, 6, 7
    mppi.Policy.add_kernel(mppi.q_cur + 0.1, 0.0, torch.eye(7))
    mppi.Policy.add_kernel(mppi.q_cur - 0.2, 0.0, torch.eye(7))
    mppi.Policy.sample_policy()
    out = mppi.propagate()
    assert all(isinstance(x, LazyRollout) for x in out) and out[0] is mppi.all_traj and out[1] is mppi.closest_dist_all
    all_traj, dist, kval, dots, acts = out
    assert all_traj.shape == (N, H, n) and kval.shape == (N, H, 2) and len(dist) == N and all_traj.ndim == 3
    # row fetches (nothing of size N x H has crossed PCIe yet)
    i, h = torch.tensor(17), torch.tensor(3)
    rows = dict(a=all_traj[5], b=all_traj[i, h], c=all_traj[-1, -1], d=dist[i, h], e=all_traj[60:64], f=mppi.qdot[0, :], g=kval[7, 2, 1],
                nb=mppi.norm_basis[i, h], n2=mppi.normal_dirs[i, h])
    assert all(x._full is None for x in out) and not mppi._cache
    full = {k: v.clone() for k, v in mppi._fetch().items()}            # now everything, once
    assert torch.equal(rows["a"], full["all_traj"][5]) and torch.equal(rows["b"], full["all_traj"][17, 3])
    assert torch.equal(rows["c"], full["all_traj"][-1, -1]) and torch.equal(rows["d"], full["closest_dist_all"][17, 3])
    assert torch.equal(rows["e"], full["all_traj"][60:64]) and torch.equal(rows["f"], full["qdot"][0])
    assert torch.equal(rows["g"], full["kernel_val_all"][7, 2, 1]) and torch.equal(rows["n2"], full["normal"][17, 3])
    assert torch.equal(rows["nb"][:, 0], full["normal"][17, 3]) and rows["nb"].shape == (7, 7)
    # tensor look-alike: functions, arithmetic, comparisons, methods, numpy
    assert torch.equal(torch.isfinite(all_traj), torch.isfinite(full["all_traj"]))
    assert torch.equal((dist < 0.3) & (dots < 0.5), (full["closest_dist_all"] < 0.3) & (full["dot_products"] < 0.5))
    assert torch.equal(all_traj[(dist < 0.3) & (dots < 0.5)], full["all_traj"][(full["closest_dist_all"] < 0.3) & (full["dot_products"] < 0.5)])
    assert torch.equal(all_traj - 1.0, full["all_traj"] - 1.0) and torch.equal(2.0 * acts, 2.0 * full["kernel_activations"])
    assert np.array_equal(np.asarray(dist), full["closest_dist_all"].numpy()) and np.array_equal(kval.numpy(), full["kernel_val_all"].numpy())
    assert torch.equal(all_traj.view(-1, n), full["all_traj"].view(-1, n)) and float(dist.min()) == float(full["closest_dist_all"].min())
    assert torch.equal(torch.cat((all_traj[:2], all_traj[2:4])), full["all_traj"][:4])
    # writes and pickles materialise
    mppi.Policy.sample_policy()
    w = mppi.propagate()
    full_w = mppi._fetch()["all_traj"].clone()
    w[0][3] = 0.0
    w[1].__iadd__(1.0)
    assert torch.equal(w[0][3], torch.zeros(H, n)) and torch.equal(w[0][4], full_w[4]) and not mppi._is_device_copy(w[0])
    back = pickle.loads(pickle.dumps(w[3]))
    assert isinstance(back, torch.Tensor) and torch.equal(back, w[3].tensor())
    # the eager form on request
    mppi.Policy.sample_policy()
    eager = mppi.propagate(fetch=True)
    assert all(isinstance(x, torch.Tensor) for x in eager)
    # a tensor nobody read before the next propagate: refuses; one that was read keeps its numbers
    mppi.Policy.sample_policy()
    stale = mppi.propagate()
    kept = stale[1].tensor().clone()
    mppi.Policy.sample_policy()
    mppi.propagate()
    with pytest.raises(RuntimeError, match="earlier propagate"):
        stale[0][0]
    assert torch.equal(stale[1].tensor(), kept) and torch.equal(stale[1][3], kept[3])

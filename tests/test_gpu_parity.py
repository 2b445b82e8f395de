"""GPU parity tests (run on a real MI355X: ``pytest -m gpu``).  Everything goes through the C-ABI
(libomds_hip.so via ctypes); the numpy oracle and the committed golden vectors are the checkers.

Tolerances: 1e-5 relative fp32 on modulated velocities / next states (BASELINE.json north_star),
teacher-forced per step so that rounding differences cannot compound; indices exact except at
(near-)ties, where the selected distances must agree instead."""
import numpy as np
import pytest

from helpers import (log_plain_bar, plain_bar, MLP_KINDS, OWN, RTOL, SCENARIOS, SEDS_FILES, assert_close, assert_velocity_plain, load, rel_err,
                     seds_of, weights_path)
from oracle import omds_oracle as orc

pytestmark = pytest.mark.gpu


def _engine(fx, H=None, N=None, flags=0):
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path(str(fx["kind"])))
    n = int(fx["q0"].shape[0])
    eng = Engine(n, int(N or fx["N"]), int(H or fx["H"]), int(fx["k"]), max_obs=max(64, 2 * fx["obs"].shape[0]), flags=flags)
    eng.set_mlp(m.W, m.b, act=m.act, skip_after=m.skip_after)
    eng.set_obstacles(fx["obs"])
    p = eng.params
    p.dt = float(fx["dt"]); p.dst_thr = float(fx["dst_thr"]); p.lin_thr = float(fx["lin_thr"]); p.rbf_p = float(fx["p"])
    mask = 0
    for l in fx["ignored_links"]:
        mask |= 1 << int(l)
    p.ignored_links = mask
    eng.push_params()
    if seds_of(fx) is not None:
        eng.set_ds_seds(fx["qf"], **seds_of(fx))
    else:
        eng.set_ds(fx["qf"])
    eng.set_cost(fx["dh_params"], fx["cost_q_min"], fx["cost_q_max"])
    return eng, m


@pytest.mark.parametrize("kind", MLP_KINDS)
def test_mlp_forward_and_vjp(kind):
    from optimalmodulationds_amd.engine import Engine
    fx = load("mlp_" + kind)
    m = orc.Mlp.from_npz(weights_path(kind))
    n = fx["x"].shape[1] - 3
    eng = Engine(n, 128, 1, 1, max_obs=8)
    eng.set_mlp(m.W, m.b, act=m.act, skip_after=m.skip_after)
    y, g, mi = eng.mlp_forward_vjp(fx["x"])
    assert_close(y, fx["y"], RTOL, "mlp forward vs reference")
    assert (mi == fx["min_idx"]).all()
    safe = (fx["min_abs_preact"] > 1e-4) | (m.act != "relu")   # rows whose ReLU masks cannot flip under fp32 rounding
    assert_close(g[safe], fx["grad"][safe], 2e-5, "vjp grad vs reference", floor=float(np.abs(fx["grad"]).max()))
    # unsafe rows: still bounded (a flipped unit changes the gradient by one weight-path, not arbitrarily)
    assert rel_err(g, fx["grad"], floor=float(np.abs(fx["grad"]).max())) < 5e-2
    if m.act == "relu":   # the reference's arithmetic restated (oracle/chain_arith.c): the device computes the oracle's BITS, forward and vjp
        oy, og, omi = orc.mlp_vjp_argmin(m, fx["x"])
        assert np.array_equal(y, oy), f"forward: {np.mean(y != oy):.4f} of the outputs are not the oracle's bits"
        assert np.array_equal(mi, omi) and np.array_equal(g, og), f"vjp: {np.mean(g != og):.4f} of the gradient entries are not the oracle's bits"
    eng.close()


@pytest.mark.parametrize("name", SCENARIOS)
def test_dist_grad_stages(name):
    fx = load(name)
    eng, m = _engine(fx)
    d, g, mind, idx = eng.dist_grad(fx["st_q"], want_mindist=True, want_idx=True)
    assert_close(mind, fx["st_mindist"], RTOL, "pass-1 min-distance matrix", floor=OWN)
    same = idx == fx["st_sort_idx"]
    if not same.all():
        picked = np.take_along_axis(mind, idx.astype(np.int64), axis=1)
        assert_close(picked, fx["st_sort_dist"], RTOL, "sorted distances at differing indices")
    assert (np.diff(np.take_along_axis(mind, idx.astype(np.int64), axis=1), axis=1) >= 0).all(), "top-k not ascending"
    assert_close(d, fx["st_distance"], 2e-5, "distance", floor=OWN)
    assert_close(g, fx["st_nn_grad"], 1e-4, "blended gradient", floor=float(np.abs(fx["st_nn_grad"]).max()))
    if m.act == "relu":   # against the oracle: the pass-1 matrix, the selection and the selected distance bit for bit
        od, og, omind, oidx = orc.distance_repulsion_nn(m, fx["st_q"], fx["obs"], int(fx["k"]), fx["ignored_links"])
        assert np.array_equal(mind, omind) and np.array_equal(d, od), "pass-1 matrix / distance: not the oracle's bits"
        assert np.array_equal(idx, oidx) or np.allclose(fx["obs"][idx], fx["obs"][oidx])
        assert_close(g, og, 1e-6, "blended gradient vs the oracle", floor=float(np.abs(og).max()))
    eng.close()


# rows of the reference's OWN steps (q_next - q) / dt that the ORACLE misses at the plain bar (tests/test_oracle_golden.py:
# ORACLE_PLAIN_MISSES -- planar 7-DoF inputs where SLEEF's sine is an ulp off MKL's closed one): the device may miss those and no others
ORACLE_PLAIN_MISSES = {"planar7_K4": 8, "planar7_128_K3": 3}


@pytest.mark.parametrize("name", SCENARIOS)
def test_teacher_forced_steps(name):
    _check_teacher_forced(name, 0)


def _check_teacher_forced(name, flags):
    """Every horizon step restarted from the reference's own state (H=1, per-rollout starts) and checked in three stages, EVERY
    row, nothing admitted (no distance envelope, no alternative ReLU-mask assignments: the device evaluates the network in the
    reference's arithmetic, oracle/chain_arith.c):
      A  network:    ReLU networks: distance BIT-IDENTICAL to the oracle's, blended gradient to 1e-6 (its softmax weights go
                     through expf); tanh networks (the device's tanhf is not numpy's): 1e-5 / 2e-5
      B  modulation: GPU step outputs vs oracle.modulation_step fed the GPU's own (distance, gradient) -- isolates the
                     per-rollout kernel                                        (1e-5)
      C  end to end: the modulated velocity at north_star's plain 1e-5 against the oracle's own step (every row), against the
                     reference's qdot (every row) and against the reference's trajectory steps (q_next - q) / dt (every row but
                     the ORACLE_PLAIN_MISSES the oracle itself misses)"""
    fx = load(name)
    eng, m = _engine(fx, H=1, flags=flags)
    H, k, N = int(fx["H"]), int(fx["k"]), int(fx["N"])
    dt = np.float32(fx["dt"])
    relu = m.act == "relu"
    prm = orc.Params(dst_thr=float(fx["dst_thr"]), lin_thr=float(fx["lin_thr"]), p=int(fx["p"]), seds=seds_of(fx))
    acc = {key: dict(rows=0, plain=0, envelope=0, mask=0, worst_plain=0.0, worst=0.0) for key in ("reference", "ref. dq/dt", "oracle")}

    def add(key, c):
        for k2 in ("rows", "plain", "envelope", "mask"):
            acc[key][k2] += c[k2]
        for k2 in ("worst_plain", "worst"):
            acc[key][k2] = max(acc[key][k2], c[k2])

    for it in range(int(fx["n_iter"])):
        pre = f"it{it}_"
        ref = fx[pre + "all_traj"]
        mu, sg, al = fx[pre + "mu_tmp"], fx[pre + "sigma_tmp"], fx[pre + "alpha_tmp"]
        eng.set_policy_samples(mu, sg, al)
        for i in range(1, H + 1):
            q = np.ascontiguousarray(ref[:, i - 1, :])
            eng.propagate(q)
            r = eng.get_rollouts()
            d_gpu, g_gpu, _, idx = eng.dist_grad(q, want_idx=True)
            d_orc, g_orc, mind_orc, oidx = orc.distance_repulsion_nn(m, q, fx["obs"], k, fx["ignored_links"])
            same_idx = (idx == oidx).all(axis=1)
            # identical dummy obstacles may swap places in the sort: harmless, identical rows
            assert same_idx.all() or np.allclose(fx["obs"][idx[~same_idx]], fx["obs"][oidx[~same_idx]]), "closest obstacles differ"
            gscale = float(np.abs(g_orc).max())
            dscale = float(np.abs(mind_orc[mind_orc < 1e5]).max())   # the largest output of the network on this step's pairs, NOT clamped at 1.0
            # --- A: network -------------------------------------------------------------------
            if relu:
                assert np.array_equal(d_gpu, d_orc), f"A distance, step {i}: not the oracle's bits ({np.abs(d_gpu - d_orc).max():.2e})"
            else:
                assert_close(d_gpu, d_orc, RTOL, f"A distance, step {i}", floor=dscale)
            e_g = np.abs(g_gpu - g_orc).max() / gscale
            assert e_g <= (1e-6 if relu else 2e-5), f"A gradient, step {i}: {e_g:.2e}"
            # --- B: modulation kernel on identical inputs ----------------------------------------
            st = orc.modulation_step(q, fx["qf"], d_gpu, g_gpu, mu, sg, al, prm)
            edge = (np.abs(st["unorm"] - prm.norm_clamp) < 1e-5) | (np.abs(st["distance"]) < 1e-6) | \
                   (np.abs(st["ga"] - prm.goal_act_cut) < 1e-6)
            keep = ~edge
            assert keep.mean() > 0.9
            assert_close(r["closest_dist_all"][keep, 0], st["distance"][keep], 1e-6, f"B distance {i}")
            assert_close(r["normal"][keep, 0], st["ghat"][keep], RTOL, f"B normal {i}")
            assert_close(r["dot_products"][keep, 0], st["dot"][keep], RTOL, f"B dot {i}")
            assert_close(r["kernel_activations"][keep, 0], st["act"][keep], RTOL, f"B act {i}")
            assert_close(r["kernel_val_all"][keep, 0], st["phi"][keep], RTOL, f"B rbf {i}")
            assert_close(r["qdot"][keep], st["u"][keep], RTOL, f"B modulated velocity {i}", floor=OWN)
            # --- C: end to end, the PLAIN bar (helpers.plain_bar), every row ------------------------
            add("oracle", plain_bar(r["qdot"], orc.modulation_step(q, fx["qf"], d_orc, g_orc, mu, sg, al, prm)["u"])[0])
            if i == 1:
                add("reference", plain_bar(r["qdot"], fx[pre + "qdot"])[0])
            if i < H and float(dt) >= 0.1:   # the reference's own step out of this state, recovered from its trajectory: (q_next - q) / dt
                add("ref. dq/dt", plain_bar(r["qdot"], (ref[:, i, :] - q) / dt)[0])   # loses ulp(q) / dt ~ 1e-6 (the integrator fixtures' dt = 0.01: 2e-5, skipped)
            assert_close(r["closest_dist_all"][:, 0], fx[pre + "closest_dist_all"][:, i - 1], RTOL, f"C distance {i}", floor=dscale)
            assert_close(r["normal"][:, 0], fx[pre + "norm_basis_n"][:, i - 1], 2e-5, f"C normal {i}")
            assert_close(r["dot_products"][:, 0], fx[pre + "dot_products"][:, i - 1], 2e-5, f"C dot {i}")
            assert_close(r["kernel_val_all"][:, 0], fx[pre + "kernel_val_all"][:, i - 1], RTOL, f"C rbf {i}")
    for key, c in acc.items():
        print(f"{name}: plain 1e-5 bar vs the {key}: {c['plain']} of {c['rows']} rows ({100.0 * c['plain'] / max(c['rows'], 1):.3f} %), worst row {c['worst']:.2e}")
        if flags == 0:
            log_plain_bar(name.split("_")[0], "teacher-forced steps", key, c)
    assert acc["oracle"]["plain"] == acc["oracle"]["rows"], acc["oracle"]
    assert acc["reference"]["plain"] == acc["reference"]["rows"], acc["reference"]
    assert acc["ref. dq/dt"]["rows"] - acc["ref. dq/dt"]["plain"] <= ORACLE_PLAIN_MISSES.get(name, 0), acc["ref. dq/dt"]
    eng.close()


@pytest.mark.parametrize("name", SCENARIOS)
def test_free_running_cost_update(name):
    _check_free_running(name, 0)


def _check_free_running(name, flags):
    fx = load(name)
    eng, m = _engine(fx, flags=flags)
    N, H, K = int(fx["N"]), int(fx["H"]), int(fx["K"])
    for it in range(int(fx["n_iter"])):
        pre = f"it{it}_"
        eng.set_policy_samples(fx[pre + "mu_tmp"], fx[pre + "sigma_tmp"], fx[pre + "alpha_tmp"])
        eng.propagate(fx[pre + "q_cur"])
        r = eng.get_rollouts()
        assert_close(r["qdot"], fx[pre + "qdot"], 5e-3, "qdot (first step; strict bar is the teacher-forced test)")
        tol = 1e-2   # free-running rollouts compound rounding differences (see test_oracle_golden)
        # A rollout may take a discrete branch of the modulation (in-collision switch, clamp, a ReLU unit) the other way at a
        # rounding-level tie and then leave the reference's trajectory for good (seen: one of 64 rollouts of franka_sub40_K4,
        # 4.8e-7 off at step 5, 7e-2 at step 6).  Both are valid fp32 evaluations; the per-step bar is the teacher-forced
        # test.  Here: such rollouts are few, they tracked the reference to rounding until they branched, and the others agree.
        e_t = np.abs(r["all_traj"] - fx[pre + "all_traj"]).max(axis=2)
        scale = max(1.0, float(np.abs(fx[pre + "all_traj"]).max()))
        off = e_t.max(axis=1) > tol * scale
        assert off.mean() <= 0.04, f"{int(off.sum())} of {N} rollouts left the reference's trajectory"
        for t in np.nonzero(off)[0]:
            first = int(np.argmax(e_t[t] > tol * scale))
            assert first >= 1 and e_t[t, first - 1] <= 2e-5 * scale, f"rollout {t} drifted instead of branching: {e_t[t]}"
        on = ~off
        assert_close(r["all_traj"][on], fx[pre + "all_traj"][on], tol, "all_traj")
        assert_close(r["closest_dist_all"][on], fx[pre + "closest_dist_all"][on], tol, "closest_dist_all")
        assert_close(r["kernel_val_all"][on], fx[pre + "kernel_val_all"][on], tol, "kernel_val_all")
        # cost: device vs oracle on the DEVICE's own rollouts (tight), and vs the reference (loose)
        cost = eng.cost()
        oc, _ = orc.evaluate_costs(r["all_traj"], r["closest_dist_all"], fx["qf"], fx["dh_params"], fx["cost_q_min"],
                                   fx["cost_q_max"])
        assert_close(cost, oc, RTOL, "cost vs oracle on device rollouts")
        # update: device reduction vs oracle arithmetic on the device's own tensors
        mu0, sg0, al0 = fx[pre + "mu_c"], fx[pre + "sigma_c"], fx[pre + "alpha_c"]
        mu, sg, al, mask, w = eng.weighted_update(float(fx["policy_upd_rate"]), float(fx["ker_thr"]), mu0, sg0, al0,
                                                  want_weights=True)
        omu, osg, oal, omask, ow = orc.shift_policy_means(cost, r["kernel_val_all"], r["kernel_activations"], mu0, sg0,
                                                          al0, fx[pre + "mu_tmp"], fx[pre + "sigma_tmp"],
                                                          fx[pre + "alpha_tmp"], float(fx["ker_thr"]),
                                                          float(fx["policy_upd_rate"]))
        assert_close(w, ow, 2e-5, "mppi weights", floor=float(ow.max()))
        assert abs(float(w.sum()) - 1.0) < 1e-5
        assert (mask == omask).all()
        assert_close(mu, omu, RTOL, "mu_c")
        assert_close(sg, osg, RTOL, "sigma_c")
        assert_close(al, oal, 2e-5, "alpha_c")
        assert_close(eng.get_qdot("weighted"), orc.get_qdot(cost, r["qdot"], "weighted"), 2e-5, "weighted qdot", floor=OWN)
        assert_close(eng.get_qdot("best"), orc.get_qdot(cost, r["qdot"], "best"), 1e-6, "best qdot")
        # against the reference's own numbers: same mask count, means close
        assert int(mask.sum()) == int(fx[pre + "n_updated"])
        # (the free-running means inherit the branching of the rollouts above: loose by nature; the strict bar is next)
        assert_close(mu, fx[pre + "mu_c_new"], 1e-3, "mu_c vs reference, free-running")
        assert_close(al, fx[pre + "alpha_c_new"], 5e-2, "alpha_c vs reference, free-running")
        # TEACHER-FORCED against the reference's own numbers: ITS rollout tensors through the device's cost kernel
        # (omds_cost_eval) and the device's update reduction (omds_weighted_update_eval) must give ITS cost, weights, mask
        # and updated means -- north_star's 1e-5, no branching involved
        c_ref = eng.cost_eval(fx[pre + "all_traj"], fx[pre + "closest_dist_all"])
        assert_close(c_ref, fx[pre + "cost"], RTOL, "cost vs reference on the reference's rollouts", floor=OWN)
        kva = fx[pre + "kernel_val_all"]
        if kva.ndim == 3 and kva.shape[2] > K:
            kva = kva[:, :, :K]
        mu_t, sg_t, al_t, mask_t, w_t = eng.weighted_update_eval(fx[pre + "cost"], kva, fx[pre + "kernel_activations"], float(fx["policy_upd_rate"]),
                                                                 float(fx["ker_thr"]), mu0, sg0, al0, want_weights=True)
        assert int(mask_t.sum()) == int(fx[pre + "n_updated"])
        if (pre + "w") in fx:
            assert_close(w_t, fx[pre + "w"], 2e-5, "weights vs reference", floor=float(fx[pre + "w"].max()))
        assert_close(mu_t, fx[pre + "mu_c_new"], RTOL, "mu_c vs reference, teacher-forced")
        assert_close(sg_t, fx[pre + "sigma_c_new"], RTOL, "sigma_c vs reference, teacher-forced")
        assert_close(al_t, fx[pre + "alpha_c_new"], RTOL, "alpha_c vs reference, teacher-forced")
    eng.close()


@pytest.mark.parametrize("name", ["franka_shelf_K6", "franka_sub40_K4", "planar7_K4", "planar2_c1_K0", "franka_at_goal_K2"])
def test_kernel_candidates(name):
    """Device-side check_traj_for_kernels vs the oracle on the device's own rollouts: same set, same
    (rollout-major) order, except states whose max RBF value sits within rounding of the threshold."""
    fx = load(name)
    eng, m = _engine(fx)
    K = int(fx["K"])
    eng.set_policy_samples(fx["it0_mu_tmp"], fx["it0_sigma_tmp"], fx["it0_alpha_tmp"])
    eng.propagate(fx["it0_q_cur"])
    r = eng.get_rollouts()
    for thr_dist, thr_kernel, thr_dot in ((0.02, 0.3, -0.9), (0.25, 0.05, -0.5), (10.0, 2.0, 2.0), (-10.0, 0.3, -0.9)):
        q, th, total = eng.kernel_candidates(thr_dist, thr_kernel, thr_dot, fx["it0_mu_c"], fx["it0_sigma_c"], K)
        want = orc.check_traj_for_kernels(r["all_traj"], r["closest_dist_all"], r["dot_products"], fx["it0_mu_c"],
                                          fx["it0_sigma_c"], thr_dist, thr_kernel, thr_dot, int(fx["p"]))
        assert total == q.shape[0] == th.shape[0]
        assert np.array_equal(q, r["all_traj"][th[:, 0], th[:, 1]])                     # gather is consistent
        assert (np.diff(th[:, 0] * 10000 + th[:, 1]) > 0).all()                          # reference order
        if q.shape == want.shape:
            assert np.array_equal(q, want)
        else:   # only threshold-straddling states may differ
            a = {tuple(x) for x in q.round(6).tolist()}; b = {tuple(x) for x in want.round(6).tolist()}
            assert len(a ^ b) <= max(2, 0.01 * len(b))
    # capacity smaller than the number found: count still reports the total
    q, th, total = eng.kernel_candidates(10.0, 2.0, 2.0, fx["it0_mu_c"], fx["it0_sigma_c"], K, cap=5)
    assert q.shape[0] == min(5, total) and total == int((~np.isnan(r["dot_products"])).sum())
    eng.close()


def test_device_sampling_statistics():
    from optimalmodulationds_amd.engine import Engine
    N, n, K = 4096, 7, 6
    eng = Engine(n, N, 2, 1, max_obs=8)
    rng = np.random.RandomState(3)
    mu_c = rng.standard_normal((K, n)).astype(np.float32)
    sg_c = (1 + rng.rand(K)).astype(np.float32)
    al_c = rng.standard_normal((K, n)).astype(np.float32)
    eng.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=42, rollout_offset=0)
    mu, sg, al = eng.get_policy_samples()
    assert np.abs(mu - mu_c[None]).max() == 0 and np.abs(sg - sg_c[None]).max() == 0   # mu_s = sigma_s = 0
    assert np.abs(al[0] - al_c).max() == 0                                              # rollout 0 = mean policy
    z = (al[1:] - al_c[None]) / 3.0
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    assert abs(float(np.mean(z ** 3))) < 0.05 and abs(float(np.mean(z ** 4)) - 3.0) < 0.15
    # a different shard (rollout_offset) draws different numbers; same seed + offset reproduces
    eng.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=42, rollout_offset=N)
    _, _, al2 = eng.get_policy_samples()
    assert np.abs(al2[1:] - al[1:]).mean() > 0.5 and np.abs(al2[0] - al_c).max() > 0
    eng.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=42, rollout_offset=0)
    _, _, al3 = eng.get_policy_samples()
    assert (al3 == al).all()
    eng.close()


def test_error_paths():
    from optimalmodulationds_amd import _lib as L
    from optimalmodulationds_amd.engine import Engine
    with pytest.raises(L.OmdsError):
        Engine(9, 8, 2, 1, max_obs=8)                       # n_dof > 7
    eng = Engine(7, 8, 2, 2, max_obs=8)
    with pytest.raises(L.OmdsError, match="network not set"):
        eng.dist_grad(np.zeros((4, 7), np.float32))
    m = orc.Mlp.from_npz(weights_path("franka"))
    with pytest.raises(L.OmdsError, match="above 4096"):     # (256 < width <= 4096 runs on the unfused GEMM path: tests/test_gpu_wide.py)
        eng.set_mlp([np.zeros((5000, 30), np.float32), np.zeros((9, 5000), np.float32)], [np.zeros(5000, np.float32), np.zeros(9, np.float32)])
    eng.set_mlp(m.W, m.b)
    eng.set_obstacles(np.zeros((9, 4), np.float32))         # beyond max_obs = 8: the context grows its obstacle buffers
    assert eng.max_obs >= 9
    with pytest.raises(L.OmdsError, match="n_closest"):
        eng.set_obstacles(np.zeros((1, 4), np.float32))     # fewer obstacles than k
    eng.set_obstacles(np.ones((4, 4), np.float32))
    with pytest.raises(L.OmdsError, match="DS not set"):
        eng.propagate(np.zeros(7, np.float32))
    eng.set_ds(np.zeros(7, np.float32))
    eng.propagate(np.ones(7, np.float32))
    with pytest.raises(L.OmdsError, match="omds_set_cost"):
        eng.cost()
    eng.close()


@pytest.mark.parametrize("name", ["franka_shelf_K6", "planar2_c1_K3", "franka_tanh_shelf_K4"])
def test_unfused_step_path(name):
    """The five-kernel step (k_pass1 / k_topk / k_pass2 / k_modulate / k_rollout_layer1; the generic path for n_dof other
    than 2 and 7), selected per context with OMDS_FLAG_UNFUSED_STEP, passes the same staged parity checks."""
    from optimalmodulationds_amd._lib import FLAG_UNFUSED_STEP
    _check_teacher_forced(name, FLAG_UNFUSED_STEP)
    _check_free_running(name, FLAG_UNFUSED_STEP)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # (network, n_dof, N, H, k, O, K, rbf_p, params overrides) -- parameter combinations no fixture holds: RBF norm orders
    # other than 2 (policy.py:41), every constant of the params struct moved off its default at once, n = 2 with k > 1.
    dict(net="franka", n=7, N=37, H=3, k=4, O=23, K=3, p=1.0, over={}),
    dict(net="franka", n=7, N=20, H=2, k=6, O=12, K=5, p=3.0, over=dict(goal_act_cut=0.2, norm_clamp=0.3, coll_slow=0.3, coll_repulse=0.2,
                                                                      softmax_k=-4.0, lvel=(0.1, 0.9, -0.6, 0.1, 20.0),
                                                                      ln=(0.0, 1.0, 0.02, 0.2, 40.0), ltau=(4.0, 1.5, 0.0, 0.2, 40.0))),
    dict(net="planar2", n=2, N=45, H=3, k=2, O=5, K=2, p=2.0, over=dict(softmax_k=-20.0)),
])
def test_parameter_space_against_oracle(case):
    """Every horizon step of a free-running device rollout is re-derived by the oracle from the device's own state:
    distance 1e-5, and (rows without a near-zero ReLU pre-activation) normal, dot product, RBF values, activation and
    the integrated velocity."""
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.engine import Engine
    n, N, H, k, O, K = (case[x] for x in ("n", "N", "H", "k", "O", "K"))
    m = orc.Mlp.from_npz(weights_path(case["net"]))
    rng = np.random.RandomState(N * 7 + k)
    if n == 7:
        obs = scenes.shelf_scene()[np.linspace(0, 293, O).astype(int)]
        q0c, qf, dst_thr, dt, ign = scenes.FRANKA_Q0, scenes.FRANKA_QF, 0.02, 0.5, [0, 1, 2]
        spread = 0.4
    else:
        obs = np.stack([rng.uniform(-6, 6, O), rng.uniform(-6, 6, O), np.zeros(O), np.full(O, 0.5)], axis=1).astype(np.float32)
        q0c, qf, dst_thr, dt, ign = np.array([-2.0, 0.5], np.float32), np.array([2.5, 0.0], np.float32), 0.25, 0.3, []
        spread = 0.8
    prm = dict(dst_thr=dst_thr, p=case["p"], **case["over"])
    eng = Engine(n, N, H, k, max_obs=max(8, O))
    eng.set_mlp(m.W, m.b, act=m.act, skip_after=m.skip_after)
    eng.set_obstacles(obs)
    P = eng.params
    P.dt, P.dst_thr, P.rbf_p = dt, dst_thr, case["p"]
    P.ignored_links = sum(1 << l for l in ign)
    for key, val in case["over"].items():
        if isinstance(val, tuple):
            getattr(P, key)[:] = val
        else:
            setattr(P, key, val)
    eng.push_params()
    eng.set_ds(qf)
    q0 = (q0c + spread * rng.standard_normal((N, n))).astype(np.float32)
    mu = (q0c + spread * rng.standard_normal((N, K, n))).astype(np.float32)
    sg = rng.uniform(0.3, 1.5, (N, K)).astype(np.float32)
    al = rng.standard_normal((N, K, n)).astype(np.float32)
    eng.set_policy_samples(mu, sg, al)
    eng.propagate(q0)
    r = eng.get_rollouts()
    eng.close()
    oprm = orc.Params(lin_thr=0.015, **{("p" if a == "p" else a): b for a, b in prm.items()})
    checked = 0
    for h in range(H):
        q = r["all_traj"][:, h]
        d, g, _, idx = orc.distance_repulsion_nn(m, q, obs, k, ign, oprm.softmax_k)
        assert_close(r["closest_dist_all"][:, h], d - np.float32(dst_thr), RTOL, f"distance h={h}", floor=OWN)
        st = orc.modulation_step(q, qf, d, g, mu, sg, al, oprm)
        checked += N    # every row: the device's ReLU masks are the oracle's
        assert_close(r["normal"][:, h], st["ghat"], 2e-5, f"normal h={h}")
        assert_close(r["dot_products"][:, h], st["dot"], 2e-5, f"dot h={h}")
        assert_close(r["kernel_val_all"][:, h], st["phi"], 2e-5, f"rbf (p={case['p']}) h={h}")
        assert_close(r["kernel_activations"][:, h], st["act"], 2e-5, f"activation h={h}")
        if h + 1 < H:
            vel = (r["all_traj"][:, h + 1] - q) / np.float32(dt)
            assert_velocity_plain(vel, q, qf, d, g, mu, sg, al, oprm, f"velocity h={h}", pad=4e-6 * max(1.0, float(np.abs(q).max())) / float(dt))
    assert checked == N * H


@pytest.mark.parametrize("name,kind,flags", [("seds_left10", "franka", 0), ("seds_sine10", "franka", 1), ("seds_right", "franka", 0),
                                              ("seds_2d", "planar2", 0)])
def test_seds_nominal_ds_on_device(name, kind, flags):
    """SEDS nominal DS (omds_set_ds_seds) inside the step kernels: one H = 1 propagate from the fixture's states (near the goal,
    near the mixture components, far away with and without the linear fallback), checked against oracle.modulation_step fed
    the device's own (distance, gradient) -- the oracle's seds_velocity is pinned to the reference's outputs on these very
    states (test_oracle_golden.py::test_seds_velocity).  A context with a SEDS nominal DS runs the step of stand-alone kernels
    (k_modulate carries the branch), with or without OMDS_FLAG_UNFUSED_STEP (flags)."""
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.engine import Engine
    fx = load(name)
    m = orc.Mlp.from_npz(weights_path(kind))
    n = fx["x"].shape[1]
    obs = scenes.shelf_scene() if n == 7 else scenes.planar2_scene(2)
    N, k = fx["x"].shape[0], (5 if n == 7 else 2)
    eng = Engine(n, N, 1, k, max_obs=512, flags=flags)
    eng.set_mlp(m.W, m.b)
    eng.set_obstacles(obs)
    eng.params.dt, eng.params.dst_thr = 0.01, 0.01
    eng.push_params()
    seds = dict(mu_in=fx["mu_in"], b=fx["b"], sigma_inv=fx["sigma_inv"], A=fx["A"], prior=fx["prior"], den=fx["den"],
                lin_thr=float(fx["lin_thr"]), seds_thr=float(fx["seds_thr"]))
    qf = fx["xT"].reshape(-1)
    eng.set_ds_seds(qf, **seds)
    rng = np.random.RandomState(2)
    K = 3
    mu_c = (qf + 0.3 * rng.standard_normal((K, n))).astype(np.float32)
    eng.sample_policy(mu_c, np.ones(K, np.float32), rng.standard_normal((K, n)).astype(np.float32), 0.0, 0.0, 0.5, K, seed=4)
    mu, sg, al = eng.get_policy_samples()
    q = np.ascontiguousarray(fx["x"])
    eng.propagate(q)                      # 2-D array: per-rollout start states
    r = eng.get_rollouts()
    d, g, _, _ = eng.dist_grad(q, want_idx=True)
    prm = orc.Params(dst_thr=0.01, seds=seds)
    st = orc.modulation_step(q, qf, d, g, mu, sg, al, prm)
    # the nominal velocity itself, through the normal . nominal-direction output and the integrated step
    v = orc.seds_velocity(q, qf, **seds)
    # states where the mixture's output cancels digits (b_j + A_j (x - mu_j) near the goal): the fp32 result depends on the
    # summation order there -- evaluate the oracle with the components and the coordinates in reversed order and set aside
    # the states where the two fp32 results disagree
    rev = {kk: (vv[::-1].copy() if isinstance(vv, np.ndarray) else vv) for kk, vv in seds.items()}
    P = np.arange(n)[::-1]
    rev.update(mu_in=rev["mu_in"][:, P], b=rev["b"][:, P], sigma_inv=rev["sigma_inv"][:, P][:, :, P], A=rev["A"][:, P][:, :, P])
    v_rev = orc.seds_velocity(q[:, P], qf[P], **rev)[:, P]
    cancel = np.abs(v - v_rev).max(axis=1) > 5e-6 * np.maximum(1.0, np.abs(v).max(axis=1))
    # the mixture's raw output within a factor 2 of seds_thr: which side of the linear-fallback switch a state lands on is decided
    # by responsibilities that are themselves ratios of underflowing exponentials
    yraw = np.linalg.norm(orc.seds_velocity(q, qf, **{**seds, "lin_thr": 1e30}), axis=1)
    near_thr = (yraw > 0.5 * seds["seds_thr"]) & (yraw < 2.0 * seds["seds_thr"])
    # every exponential of the state in the denormal range (or zero): whether the one surviving component is the smallest
    # denormal or 0 -- mixture output or linear fallback -- hangs on the last bit of its Mahalanobis form
    ddm = (q - qf)[:, None, :] - seds["mu_in"][None]
    denorm = (-0.5 * np.einsum("ngr,grc,ngc->ng", ddm, seds["sigma_inv"], ddm)).max(axis=1) < -85.0
    edge = cancel | denorm | near_thr | (np.abs(st["unorm"] - prm.norm_clamp) < 1e-4) | (np.abs(st["distance"]) < 1e-6) | (np.abs(st["ga"] - prm.goal_act_cut) < 1e-6) | \
           (np.abs(np.linalg.norm(v, axis=1) - 1e-2) < 1e-4) | ~np.isfinite(st["u"]).all(axis=1)
    keep = ~edge
    # far outside the demonstrations the responsibilities are ratios of exp(-100 .. -10^5): order-sensitive in fp32 for ~40 % of
    # such random states (the reference's own answer there is one of many); the others must agree
    assert keep.mean() > 0.4
    scale = max(1.0, float(np.abs(st["u"][keep]).max()))
    # 2e-4: the Mahalanobis forms are sums of terms ~10^4 x their result (covariances with condition numbers of 10^3 .. 10^4),
    # and their fp32 rounding is amplified by exp() into the responsibilities; the reference-captured scenario with a SEDS
    # nominal DS (franka_seds_integrator_N1) passes the generic tests at their usual bars
    assert_close(r["dot_products"][keep, 0], st["dot"][keep], 2e-4, "normal . nominal direction")
    assert np.abs(r["qdot"][keep] - st["u"][keep]).max() <= 2e-4 * scale, float(np.abs(r["qdot"][keep] - st["u"][keep]).max())
    # where fp32 is well conditioned -- some component's exponent above -20, so the responsibilities are ratios of normal
    # numbers -- the device must agree at the bar of every other stage
    well = keep & ((-0.5 * np.einsum("ngr,grc,ngc->ng", ddm, seds["sigma_inv"], ddm)).max(axis=1) > -20.0)
    print(f"SEDS {name}: kept {keep.mean():.2f}, well-conditioned {well.mean():.2f}, "
          f"max err kept {float(np.abs(r['qdot'][keep] - st['u'][keep]).max()):.2e}"
          + (f", well {float(np.abs(r['qdot'][well] - st['u'][well]).max()):.2e}" if well.any() else ""))
    if well.any():
        assert np.abs(r["qdot"][well] - st["u"][well]).max() <= 5e-5 * scale, float(np.abs(r["qdot"][well] - st["u"][well]).max())
    eng.close()

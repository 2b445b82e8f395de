"""Pins the CPU oracle (oracle/omds_oracle.py) to vectors captured from the real reference.

CPU-only.  Every committed fixture under tests/golden/ is checked: raw MLP forward, vjp of the
arg-min link, the pass-1 min-distance matrix and top-k indices, the full propagate() outputs,
the cost and the cost-weighted policy update."""
import numpy as np
import pytest

from helpers import OWN, SEDS_FILES, seds_of, MLP_KINDS, RTOL, SCENARIOS, assert_close, load, log_plain_bar, plain_bar, weights_path
from oracle import omds_oracle as orc


@pytest.mark.parametrize("kind", MLP_KINDS)
def test_mlp_forward_and_vjp(kind):
    fx = load("mlp_" + kind)
    m = orc.Mlp.from_npz(weights_path(kind))
    y = orc.mlp_forward(m, fx["x"])
    assert_close(y, fx["y"], RTOL, "mlp forward")
    y2, grad, mi = orc.mlp_vjp_argmin(m, fx["x"])
    assert_close(y2, fx["y_vjp"], RTOL, "vjp forward")
    assert (mi == fx["min_idx"]).all()
    # rows whose smallest |pre-activation| is within fp32 rounding of 0 may flip a ReLU mask
    safe = (fx["min_abs_preact"] > 1e-4) | (m.act != "relu")
    assert safe.sum() > 0.8 * len(safe)
    assert_close(grad[safe], fx["grad"][safe], 2e-5, "vjp grad", floor=float(np.abs(fx["grad"]).max()))


WGRAD_KINDS = ["franka", "planar7", "franka_tanh"]


@pytest.mark.parametrize("kind", WGRAD_KINDS)
def test_jacobian_columns_and_closest_gradient(kind):
    """compute_signed_distance_wgrad / _wgrad2 / dist_grad_closest of the reference (tools/make_golden_wgrad.py) vs the oracle."""
    fx = load("wgrad_" + kind)
    m = orc.Mlp.from_npz(weights_path(kind))
    x = fx["x"]
    C = m.W[-1].shape[0]
    safe = (fx["min_abs_preact"] > 1e-4) | (m.act != "relu")
    assert safe.sum() > 0.8 * len(safe)
    floor = float(np.abs(fx["all_grads"]).max())
    d, J = orc.mlp_jacobian(m, x, list(range(C)))
    assert_close(d, fx["all_dist"], RTOL, "distances")
    assert_close(J[safe], fx["all_grads"][safe], 2e-5, "all Jacobian columns", floor=floor)
    _, J = orc.mlp_jacobian(m, x, fx["cols"])
    assert_close(J[safe], fx["cols_grads"][safe], 2e-5, "listed Jacobian columns", floor=floor)
    d, g, mi = orc.mlp_closest_wgrad(m, x)
    assert (mi == fx["closest_idx"]).all()
    assert_close(g[safe], fx["closest_grads"][safe], 2e-5, "closest gradient", floor=floor)
    if "w2_grads" in fx:
        assert (mi == fx["w2_idx"]).all()
        assert_close(g[safe, :, 0], fx["w2_grads"][safe], 2e-5, "wgrad2 gradient", floor=floor)
    nb = fx["dgc_dist"].shape[0]                                   # dist_grad_closest evaluated maxInputSize rows only
    assert (mi[:nb] == fx["dgc_idx"]).all()
    assert_close(g[:nb][safe[:nb]], fx["dgc_grads"][safe[:nb]], 2e-5, "dist_grad_closest gradient", floor=floor)
    # a permuted link order: columns, arg-min and Jacobian columns follow the re-ordered outputs
    d, g, mi = orc.mlp_closest_wgrad(m, x, order=fx["order"])
    assert_close(d, fx["ord_closest_dist"], RTOL, "re-ordered distances")
    assert (mi == fx["ord_closest_idx"]).all()
    assert_close(g[safe], fx["ord_closest_grads"][safe], 2e-5, "closest gradient, re-ordered", floor=floor)
    _, J = orc.mlp_jacobian(m, x, fx["cols"], order=fx["order"])
    assert_close(J[safe], fx["ord_cols_grads"][safe], 2e-5, "listed columns, re-ordered", floor=floor)


def _model(fx):
    return orc.Mlp.from_npz(weights_path(str(fx["kind"])))


@pytest.mark.parametrize("name", SCENARIOS)
def test_stage_intermediates(name):
    fx = load(name)
    m = _model(fx)
    d, g, mind, sidx = orc.distance_repulsion_nn(m, fx["st_q"], fx["obs"], int(fx["k"]), fx["ignored_links"])
    assert_close(mind, fx["st_mindist"], RTOL, "pass-1 min-distance matrix")
    same = (sidx == fx["st_sort_idx"])
    if not same.all():  # ties / near-ties only: the sorted distances must still agree
        picked = np.take_along_axis(mind, sidx, axis=1)
        assert_close(picked, fx["st_sort_dist"], RTOL, "sorted distances at differing indices")
    assert_close(d, fx["st_distance"], 2e-5, "distance")
    assert_close(g, fx["st_nn_grad"], 1e-4, "blended gradient", floor=float(np.abs(fx["st_nn_grad"]).max()))


def _prm(fx, basis=False):
    return orc.Params(dst_thr=float(fx["dst_thr"]), lin_thr=float(fx["lin_thr"]), p=int(fx["p"]), want_basis=basis, seds=seds_of(fx))


@pytest.mark.parametrize("name", SCENARIOS)
def test_teacher_forced_steps(name):
    """Strict per-step parity: every horizon step is restarted from the REFERENCE's state
    all_traj[:, i-1] (an H=1 propagate with per-rollout start states), so rounding differences
    cannot compound through the steep sigmoids / ReLU masks of later steps.  Bar: 1e-5 rel."""
    fx = load(name)
    m = _model(fx)
    N, H, k = int(fx["N"]), int(fx["H"]), int(fx["k"])
    dt = np.float32(fx["dt"])
    for it in range(int(fx["n_iter"])):
        pre = f"it{it}_"
        ref = fx[pre + "all_traj"]
        for i in range(1, H + 1):
            out = orc.propagate(m, ref[:, i - 1, :], fx["qf"], fx["obs"], N=N, H=1, dt=float(dt), k=k,
                                ignored_links=fx["ignored_links"], mu_tmp=fx[pre + "mu_tmp"],
                                sigma_tmp=fx[pre + "sigma_tmp"], alpha_tmp=fx[pre + "alpha_tmp"], prm=_prm(fx))
            if i < H:
                assert_close(ref[:, i - 1, :] + dt * out.qdot, ref[:, i, :], RTOL, f"next state, step {i}")
            if i == 1:   # the velocity at its OWN scale (never clamped up to 1): the bar the device is held to (helpers.plain_bar)
                assert_close(out.qdot, fx[pre + "qdot"], RTOL, "qdot (modulated velocity)", floor=OWN)
            assert_close(out.closest_dist_all[:, 0], fx[pre + "closest_dist_all"][:, i - 1], RTOL, f"distance {i}")
            assert_close(out.dot_products[:, 0], fx[pre + "dot_products"][:, i - 1], RTOL, f"dot {i}")
            assert_close(out.kernel_activations[:, 0], fx[pre + "kernel_activations"][:, i - 1], 2e-5, f"act {i}")
            assert_close(out.kernel_val_all[:, 0], fx[pre + "kernel_val_all"][:, i - 1], RTOL, f"rbf {i}")
            assert_close(out.norm_basis_n[:, 0], fx[pre + "norm_basis_n"][:, i - 1], 2e-5, f"normal {i}")


# Plain north-star bar of the ORACLE against the reference's own steps, every row, no envelope, no mask alternatives
# (helpers.plain_bar: |u - u_ref| <= 1e-5 max|u_ref|): the comparator of the device's table (tests/conftest.py prints both).  With the
# reference's arithmetic restated (oracle/chain_arith.c) every row of every Franka and planar 2-DoF fixture meets it; the rows of
# the planar 7-DoF fixtures that do not (network outputs ~10 with an ulp of 1e-6, times sigmoid slopes of 100) sit on inputs
# where SLEEF's sine is an ulp off MKL's closed one.  Floors = what was measured, not a margin.
ORACLE_PLAIN_MISSES = {"planar7_K4": 8, "planar7_128_K3": 3}


@pytest.mark.parametrize("name", SCENARIOS)
def test_oracle_meets_the_plain_bar(name):
    fx = load(name)
    m = _model(fx)
    N, H, k = int(fx["N"]), int(fx["H"]), int(fx["k"])
    dt = np.float32(fx["dt"])
    acc = {key: dict(rows=0, plain=0, envelope=0, mask=0, worst_plain=0.0, worst=0.0) for key in ("reference", "ref. dq/dt")}
    for it in range(int(fx["n_iter"])):
        pre = f"it{it}_"
        ref = fx[pre + "all_traj"]
        for i in range(1, H + 1):
            q = np.ascontiguousarray(ref[:, i - 1, :])
            out = orc.propagate(m, q, fx["qf"], fx["obs"], N=N, H=1, dt=float(dt), k=k, ignored_links=fx["ignored_links"],
                                mu_tmp=fx[pre + "mu_tmp"], sigma_tmp=fx[pre + "sigma_tmp"], alpha_tmp=fx[pre + "alpha_tmp"], prm=_prm(fx))
            pairs = []
            if i == 1:
                pairs.append(("reference", fx[pre + "qdot"]))
            if i < H and float(dt) >= 0.1:      # (q_next - q) / dt loses ulp(q) / dt; the integrator fixtures' dt = 0.01 is skipped
                pairs.append(("ref. dq/dt", (ref[:, i, :] - q) / dt))
            for key, u_ref in pairs:
                c = plain_bar(out.qdot, u_ref, np.zeros(N, bool))[0]
                for k2 in ("rows", "plain", "envelope", "mask"):
                    acc[key][k2] += c[k2]
                acc[key]["worst"] = max(acc[key]["worst"], c["worst"])
    for key, c in acc.items():
        log_plain_bar(name.split("_")[0], "ORACLE teacher-forced steps", key, c)
    assert acc["reference"]["plain"] == acc["reference"]["rows"], acc["reference"]
    assert acc["reference"]["worst"] <= 1e-6, acc["reference"]
    miss = acc["ref. dq/dt"]["rows"] - acc["ref. dq/dt"]["plain"]
    assert miss <= ORACLE_PLAIN_MISSES.get(name, 0), acc["ref. dq/dt"]
    if name not in ORACLE_PLAIN_MISSES:
        assert acc["ref. dq/dt"]["worst"] <= 5e-6, acc["ref. dq/dt"]


@pytest.mark.parametrize("name", SCENARIOS)
def test_propagate_cost_update(name):
    fx = load(name)
    m = _model(fx)
    N, H, k, K = int(fx["N"]), int(fx["H"]), int(fx["k"]), int(fx["K"])
    for it in range(int(fx["n_iter"])):
        pre = f"it{it}_"
        out = orc.propagate(m, fx[pre + "q_cur"], fx["qf"], fx["obs"], N=N, H=H, dt=float(fx["dt"]), k=k,
                            ignored_links=fx["ignored_links"], mu_tmp=fx[pre + "mu_tmp"],
                            sigma_tmp=fx[pre + "sigma_tmp"], alpha_tmp=fx[pre + "alpha_tmp"],
                            prm=_prm(fx, pre + "norm_basis" in fx))
        # free-running rollouts compound the per-step 1e-6-class differences through k=100
        # sigmoids and ReLU-mask flips (seen up to 2e-3); the strict bar is the teacher-forced test
        tol = 1e-2
        # seeded synthetic weights (franka_skip): a random network has many more pre-activations near zero than a trained
        # one, and one flipped unit on one rollout at the last step moves its normal by 1.4e-2 (trajectories agree to 1.3e-6)
        tol_n = 2e-2 if str(fx["kind"]) == "franka_skip" else tol
        assert_close(out.qdot, fx[pre + "qdot"], RTOL, "qdot (modulated velocity)", floor=OWN)
        # a rollout may take a discrete branch (a ReLU unit, the in-collision switch) the other way at a rounding-level tie and
        # leave the reference's trajectory for good: few, and they tracked the reference until they branched
        e_t = np.abs(out.norm_basis_n - fx[pre + "norm_basis_n"]).max(axis=2)
        off = e_t.max(axis=1) > tol_n
        assert off.mean() <= 0.04, f"{int(off.sum())} of {N} rollouts left the reference's trajectory"
        on = ~off
        assert_close(out.all_traj[on], fx[pre + "all_traj"][on], tol, "all_traj")
        assert_close(out.closest_dist_all[on], fx[pre + "closest_dist_all"][on], tol, "closest_dist_all")
        assert_close(out.dot_products[on], fx[pre + "dot_products"][on], tol_n, "dot_products")
        assert_close(out.kernel_activations[on], fx[pre + "kernel_activations"][on], 2e-2, "kernel_activations")
        assert_close(out.kernel_val_all[on], fx[pre + "kernel_val_all"][on], tol, "kernel_val_all")
        assert_close(out.norm_basis_n[on], fx[pre + "norm_basis_n"][on], tol_n, "normal direction")
        if pre + "norm_basis" in fx and out.norm_basis is not None:
            # Householder completion is ill-conditioned in g[0] when |g[0]| << 1 (seen: 6e-4)
            assert_close(out.norm_basis[on], fx[pre + "norm_basis"][on], 2e-2, "full QR basis")
        # cost and update are checked on the REFERENCE's rollouts so that errors do not compound
        cost, parts = orc.evaluate_costs(fx[pre + "all_traj"], fx[pre + "closest_dist_all"], fx["qf"],
                                         fx["dh_params"], fx["cost_q_min"], fx["cost_q_max"])
        assert_close(parts["goal"], fx[pre + "cost_goal"], RTOL, "goal cost")
        assert_close(parts["fk"], fx[pre + "cost_fk"], RTOL, "fk cost")
        assert_close(cost, fx[pre + "cost"], RTOL, "total cost")
        mu, sg, al, mask, w = orc.shift_policy_means(
            fx[pre + "cost"], fx[pre + "kernel_val_all"], fx[pre + "kernel_activations"], fx[pre + "mu_c"],
            fx[pre + "sigma_c"], fx[pre + "alpha_c"], fx[pre + "mu_tmp"], fx[pre + "sigma_tmp"],
            fx[pre + "alpha_tmp"], float(fx["ker_thr"]), float(fx["policy_upd_rate"]))
        assert_close(w, fx[pre + "w"], RTOL, "mppi weights", floor=float(fx[pre + "w"].max()))
        assert int(mask.sum()) == int(fx[pre + "n_updated"])
        assert_close(mu, fx[pre + "mu_c_new"], RTOL, "mu_c")
        assert_close(sg, fx[pre + "sigma_c_new"], RTOL, "sigma_c")
        assert_close(al, fx[pre + "alpha_c_new"], RTOL, "alpha_c")
        assert_close(orc.get_qdot(fx[pre + "cost"], fx[pre + "qdot"], "weighted"), fx[pre + "qdot_weighted"],
                     RTOL, "weighted qdot")
        assert_close(orc.get_qdot(fx[pre + "cost"], fx[pre + "qdot"], "best"), fx[pre + "qdot_best"],
                     RTOL, "best qdot")


@pytest.mark.parametrize("name", SCENARIOS)
def test_kernel_candidates_and_add_kernel(name):
    """check_traj_for_kernels / add_kernel (policy.py:129-175) as captured from the reference: the oracle
    and the facade's host-side bookkeeping reproduce the candidate list (same order) and the policy
    after adding the candidate closest to q_cur."""
    import torch
    from optimalmodulationds_amd.policy import TensorPolicyMPPI
    fx = load(name)
    K, n, N = int(fx["K"]), fx["q0"].shape[0], int(fx["N"])
    thr = fx["cand_thr"]
    want = fx["cand_q"]
    got = orc.check_traj_for_kernels(fx["it0_all_traj"], fx["it0_closest_dist_all"], fx["it0_dot_products"], fx["it0_mu_c"],
                                     fx["it0_sigma_c"], thr[0], thr[1], thr[2], int(fx["p"]))
    assert got.shape == want.shape and np.array_equal(got, want)
    P = TensorPolicyMPPI(N, n)
    P.n_kernels = K
    P.mu_c[:K] = torch.from_numpy(fx["it0_mu_c"]); P.sigma_c[:K] = torch.from_numpy(fx["it0_sigma_c"])
    P.alpha_c[:K] = torch.from_numpy(fx["it0_alpha_c"]); P.p = int(fx["p"])
    c2 = P.check_traj_for_kernels(torch.from_numpy(fx["it0_all_traj"]), torch.from_numpy(fx["it0_closest_dist_all"]),
                                  torch.from_numpy(fx["it0_dot_products"]), float(thr[0]), float(thr[1]), float(thr[2]))
    assert np.array_equal(c2.numpy(), want)
    if "add_mu_c" in fx:
        ci, ti, hi = (int(x) for x in fx["add_idx"])
        P.sigma_c_nominal = float(fx["add_sigma_c"][-1])
        P.add_kernel(c2[ci], fx["it0_closest_dist_all"][ti, hi], fx["add_basis"][-1])
        assert P.n_kernels == int(fx["add_n_kernels"]) == K + 1
        assert np.array_equal(P.mu_c[:K + 1].numpy(), fx["add_mu_c"]) and np.array_equal(P.alpha_c[:K + 1].numpy(), fx["add_alpha_c"])
        assert np.array_equal(P.sigma_c[:K + 1].numpy(), fx["add_sigma_c"]) and np.array_equal(P.kernel_gammas[:K + 1].numpy(), fx["add_gammas"])


def test_torch_baseline_matches_the_numpy_oracle():
    """bench.py's CPU baseline (oracle/torch_baseline.py, the reference's unfused op sequence in torch-CPU) reproduces the
    numpy oracle -- which the tests above pin to the reference's own outputs -- on a fixture's inputs: rollouts, cost, update."""
    import torch
    from oracle.torch_baseline import TorchPlanner, _t
    fx = load("franka_sub40_K4")
    m = orc.Mlp.from_npz(weights_path("franka"))
    N, H, k, K = int(fx["N"]), int(fx["H"]), int(fx["k"]), int(fx["K"])
    prm = orc.Params(dst_thr=float(fx["dst_thr"]), lin_thr=float(fx["lin_thr"]), p=int(fx["p"]), seds=seds_of(fx))
    pl = TorchPlanner(m, fx["obs"], fx["qf"], fx["dh_params"], fx["cost_q_min"], fx["cost_q_max"], dt=float(fx["dt"]), k=k,
                      ignored_links=[int(l) for l in fx["ignored_links"]], prm=prm)
    mu, sg, al = fx["it0_mu_tmp"][:, :K], fx["it0_sigma_tmp"][:, :K], fx["it0_alpha_tmp"][:, :K]
    torch.set_num_threads(4)
    with torch.no_grad():
        traj, dist, kval, acts, qdot = pl.propagate(_t(fx["it0_q_cur"]), H, _t(mu), _t(sg), _t(al))
        cost = pl.evaluate_costs(traj, dist)
        mu_n, sg_n, al_n, w = pl.shift_policy_means(cost, kval, acts, _t(fx["it0_mu_c"][:K]), _t(fx["it0_sigma_c"][:K]),
                                                    _t(fx["it0_alpha_c"][:K]), _t(mu), _t(sg), _t(al), 0.1, float(fx["ker_thr"]))
    o = orc.propagate(m, fx["it0_q_cur"], fx["qf"], fx["obs"], N=N, H=H, dt=float(fx["dt"]), k=k,
                      ignored_links=fx["ignored_links"], mu_tmp=mu, sigma_tmp=sg, alpha_tmp=al, prm=prm)
    assert_close(qdot.numpy(), o.qdot, 2e-5, "qdot")
    assert_close(dist.numpy()[:, 0], o.closest_dist_all[:, 0], RTOL, "distance")
    assert_close(traj.numpy(), o.all_traj, 1e-2, "free-running rollouts")
    oc, _ = orc.evaluate_costs(traj.numpy(), dist.numpy(), fx["qf"], fx["dh_params"], fx["cost_q_min"], fx["cost_q_max"])
    assert_close(cost.numpy(), oc, 2e-5, "cost")
    omu, osg, oal, omask, ow = orc.shift_policy_means(cost.numpy(), kval.numpy(), acts.numpy(), fx["it0_mu_c"][:K],
                                                      fx["it0_sigma_c"][:K], fx["it0_alpha_c"][:K], mu, sg, al, 0.1, float(fx["ker_thr"]))
    assert_close(w.numpy(), ow, 1e-5, "weights", floor=float(ow.max()))
    assert_close(al_n.numpy(), oal, 2e-5, "alpha_c")


@pytest.mark.parametrize("name", SEDS_FILES)
def test_seds_velocity(name):
    """SEDS.get_velocity (SEDS.py:59-74) on the reference's shipped mixtures, one state per reference call: near the goal
    (raw mixture output), near the components, far away (normalised; linear fallback where the mixture is weak)."""
    fx = load(name)
    y = orc.seds_velocity(fx["x"], fx["xT"].reshape(-1), fx["mu_in"], fx["b"], fx["sigma_inv"], fx["A"], fx["prior"], fx["den"],
                          float(fx["lin_thr"]), float(fx["seds_thr"]))
    # relative to the largest output: the 2-D mixture (fitted in pixel units) answers ~136 near its goal, where b + A (x - mu)
    # cancels three digits -- 3e-5 between two fp32 summation orders
    assert_close(y, fx["y"], 5e-5, "SEDS velocity vs reference", floor=1.0)


def test_train_oracle_matches_the_reference_run():
    """oracle/train_oracle.py (one epoch of train_sdf.py:105-113 + torch's Adam arithmetic in numpy) against the run of the
    reference's own model class and torch.optim.Adam captured in train_sdf_planar2.npz: the loss of each of the 100 epochs and
    the weights after 10 and 100 epochs.  Weight DELTAS are compared (what training moved), relative to the largest delta."""
    from oracle import train_oracle as tro
    fx = load("train_sdf_planar2")
    nl = len([k for k in fx if k.startswith("W0_")])
    st = tro.TrainState([fx[f"W0_{i}"] for i in range(nl)], [fx[f"b0_{i}"] for i in range(nl)])
    losses = []
    for e in range(int(fx["epochs"])):
        losses.append(tro.train_step(st, fx["x"], fx["y"], lr=float(fx["lr"])))
        if e + 1 in (10, 100):
            tag = f"W{e + 1}_"
            dmax = max(float(np.abs(fx[f"{tag}{i}"] - fx[f"W0_{i}"]).max()) for i in range(nl))
            for i in range(nl):
                err = float(np.abs((st.W[i] - fx[f"W0_{i}"]) - (fx[f"{tag}{i}"] - fx[f"W0_{i}"])).max())
                assert err <= 2e-3 * dmax, (e + 1, i, err, dmax)
    rel = np.abs(np.asarray(losses) - fx["losses"]) / fx["losses"]
    assert rel.max() <= 1e-5, float(rel.max())
    assert abs(tro.mse(tro.forward(st, fx["x"])[-1], fx["y"]) - float(fx["final_eval"])) <= 1e-5 * float(fx["final_eval"])

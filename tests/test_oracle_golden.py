"""Pins the CPU oracle (oracle/omds_oracle.py) to vectors captured from the real reference.

CPU-only.  Every committed fixture under tests/golden/ is checked: raw MLP forward, vjp of the
arg-min link, the pass-1 min-distance matrix and top-k indices, the full propagate() outputs,
the cost and the cost-weighted policy update."""
import numpy as np
import pytest

from helpers import MLP_KINDS, RTOL, SCENARIOS, assert_close, load, weights_path
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
    safe = fx["min_abs_preact"] > 1e-4
    assert safe.sum() > 0.8 * len(safe)
    assert_close(grad[safe], fx["grad"][safe], 2e-5, "vjp grad", floor=float(np.abs(fx["grad"]).max()))


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


@pytest.mark.parametrize("name", SCENARIOS)
def test_propagate_cost_update(name):
    fx = load(name)
    m = _model(fx)
    N, H, k, K = int(fx["N"]), int(fx["H"]), int(fx["k"]), int(fx["K"])
    prm = orc.Params(dst_thr=float(fx["dst_thr"]), lin_thr=float(fx["lin_thr"]), p=int(fx["p"]),
                     want_basis=("it0_norm_basis" in fx))
    for it in range(int(fx["n_iter"])):
        pre = f"it{it}_"
        out = orc.propagate(m, fx[pre + "q_cur"], fx["qf"], fx["obs"], N=N, H=H, dt=float(fx["dt"]), k=k,
                            ignored_links=fx["ignored_links"], mu_tmp=fx[pre + "mu_tmp"],
                            sigma_tmp=fx[pre + "sigma_tmp"], alpha_tmp=fx[pre + "alpha_tmp"], prm=prm)
        tol = 5e-5  # a rollout integrates H steps of a 1e-5-class velocity error
        assert_close(out.all_traj, fx[pre + "all_traj"], tol, "all_traj")
        assert_close(out.qdot, fx[pre + "qdot"], RTOL * 2, "qdot (modulated velocity)")
        assert_close(out.closest_dist_all, fx[pre + "closest_dist_all"], tol, "closest_dist_all")
        assert_close(out.dot_products, fx[pre + "dot_products"], 1e-4, "dot_products")
        assert_close(out.kernel_activations, fx[pre + "kernel_activations"], 1e-4, "kernel_activations")
        assert_close(out.kernel_val_all, fx[pre + "kernel_val_all"], tol, "kernel_val_all")
        assert_close(out.norm_basis_n, fx[pre + "norm_basis_n"], 1e-4, "normal direction")
        if pre + "norm_basis" in fx and out.norm_basis is not None:
            assert_close(out.norm_basis, fx[pre + "norm_basis"], 1e-4, "full QR basis")
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

"""MPPI_toy variant (ds_mppi/functions/MPPI_toy.py + cost_toy.py): nominal DS (q - qf) @ A, the toy's
constants, kernel values stored times activation, update mask without the rollout-0 term, 3-term cost, DOF+2-input
network with planar obstacle points.  Fixtures come from the reference (tools/make_golden_toy.py).
CPU: the oracle against the fixtures.  GPU: the HIP path against fixtures and oracle, through the C-ABI and through the
facade class optimalmodulationds_amd.mppi_toy.MPPI."""
import numpy as np
import pytest

from helpers import RTOL, TOY_SCENARIOS, assert_close, load, weights_path
from oracle import omds_oracle as orc

TOY_TERMS = ("goal", "coll", "stag")


def toy_prm(fx):
    return orc.Params(dst_thr=float(fx["dst_thr"]), lin_thr=0.0, p=int(fx["p"]), lvel=(0.0, 1.0, -0.2, 0.0, 100.0),
                      ln=(0.0, 1.0, 0.0, 0.5, 30.0), ltau=(3.0, 1.0, 0.0, 0.5, 30.0), goal_act_cut=0.3, coll_repulse=0.05,
                      A=fx["A"], kval_times_act=True)


def test_fixtures_present():
    assert len(TOY_SCENARIOS) >= 3


def test_oracle_mlp_planar_points():
    fx = load("mlp_toy2")
    m = orc.Mlp.from_npz(weights_path("toy2"))
    assert m.W[0].shape[1] == 12            # 3 * (DOF + 2)
    assert_close(orc.mlp_forward(m, fx["x"]), fx["y"], RTOL, "forward")
    y, g, mi = orc.mlp_vjp_argmin(m, fx["x"])
    assert (mi == fx["min_idx"]).all()
    assert_close(g, fx["grad"], 2e-5, "vjp", floor=float(np.abs(fx["grad"]).max()))


@pytest.mark.parametrize("name", TOY_SCENARIOS)
def test_oracle_teacher_forced_and_update(name):
    fx = load(name)
    m = orc.Mlp.from_npz(weights_path("toy2"))
    N, H, k, K = int(fx["N"]), int(fx["H"]), int(fx["k"]), int(fx["K"])
    dt = np.float32(fx["dt"])
    for it in range(int(fx["n_iter"])):
        pre = f"it{it}_"
        ref = fx[pre + "all_traj"]
        for i in range(1, H + 1):
            out = orc.propagate(m, ref[:, i - 1, :], fx["qf"], fx["obs"], N=N, H=1, dt=float(dt), k=k, ignored_links=[],
                                mu_tmp=fx[pre + "mu_tmp"], sigma_tmp=fx[pre + "sigma_tmp"], alpha_tmp=fx[pre + "alpha_tmp"],
                                prm=toy_prm(fx))
            if i < H:
                assert_close(ref[:, i - 1, :] + dt * out.qdot, ref[:, i, :], RTOL, f"next state, step {i}")
            if i == 1:
                assert_close(out.qdot, fx[pre + "qdot"], RTOL, "qdot")
            assert_close(out.closest_dist_all[:, 0], fx[pre + "closest_dist_all"][:, i - 1], RTOL, f"distance {i}")
            assert_close(out.dot_products[:, 0], fx[pre + "dot_products"][:, i - 1], RTOL, f"dot {i}")
            assert_close(out.kernel_val_all[:, 0], fx[pre + "kernel_val_all"][:, i - 1], 2e-5, f"phi*act {i}")
            assert_close(out.norm_basis_n[:, 0], fx[pre + "norm_basis_n"][:, i - 1], 2e-5, f"normal {i}")
        cost, _ = orc.evaluate_costs(fx[pre + "all_traj"], fx[pre + "closest_dist_all"], fx["qf"], np.zeros((3, 4), np.float32),
                                     fx["cost_q_min"], fx["cost_q_max"], terms=TOY_TERMS)
        assert_close(cost, fx[pre + "cost"], RTOL, "3-term cost")
        mu, sg, al, mask, w = orc.shift_policy_means(fx[pre + "cost"], fx[pre + "kernel_val_all"], None, fx[pre + "mu_c"],
                                                     fx[pre + "sigma_c"], fx[pre + "alpha_c"], fx[pre + "mu_tmp"], fx[pre + "sigma_tmp"],
                                                     fx[pre + "alpha_tmp"], float(fx["ker_thr"]), float(fx["policy_upd_rate"]), toy=True)
        assert_close(w, fx[pre + "w"], RTOL, "weights", floor=float(fx[pre + "w"].max()))
        assert np.array_equal(mask, fx[pre + "mask"])
        assert_close(mu, fx[pre + "mu_c_new"], RTOL, "mu_c")
        assert_close(sg, fx[pre + "sigma_c_new"], RTOL, "sigma_c")
        assert_close(al, fx[pre + "alpha_c_new"], RTOL, "alpha_c")


# ---------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------
def _obs4(fx):
    o = np.zeros((fx["obs"].shape[0], 4), np.float32)
    o[:, :2] = fx["obs"][:, :2]
    o[:, 3] = fx["obs"][:, 2]
    return o


def _engine(fx, N, H):
    from optimalmodulationds_amd.engine import Engine
    from optimalmodulationds_amd.mppi_toy import toy_params
    z = np.load(weights_path("toy2"))
    eng = Engine(2, N, H, int(fx["k"]), max_obs=64)
    eng.set_mlp([z[f"W{i}"] for i in range(5)], [z[f"b{i}"] for i in range(5)])
    eng.set_obstacles(_obs4(fx))
    toy_params(eng.params)
    eng.params.dt = float(fx["dt"]); eng.params.dst_thr = float(fx["dst_thr"]); eng.params.ignored_links = 0
    eng.push_params()
    eng.set_ds_matrix(fx["qf"], fx["A"])
    eng.set_cost(np.zeros((3, 4), np.float32), fx["cost_q_min"], fx["cost_q_max"])
    return eng


@pytest.mark.gpu
def test_gpu_mlp_planar_points():
    from optimalmodulationds_amd.engine import Engine
    fx = load("mlp_toy2")
    z = np.load(weights_path("toy2"))
    eng = Engine(2, 128, 1, 1, max_obs=8)
    eng.set_mlp([z[f"W{i}"] for i in range(5)], [z[f"b{i}"] for i in range(5)])
    y, g, mi = eng.mlp_forward_vjp(fx["x"])
    eng.close()
    assert_close(y, fx["y"], RTOL, "forward")
    assert (mi == fx["min_idx"]).all()
    assert_close(g, fx["grad"], 2e-5, "vjp", floor=float(np.abs(fx["grad"]).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", TOY_SCENARIOS)
def test_gpu_teacher_forced_cost_update(name):
    """Per-step parity from the reference's states (tolerances as in test_gpu_parity.py: stage C end to end 2e-4 because
    the k=100 / k=30 sigmoids amplify 1e-6 distance differences), then cost and update on the device's own rollouts
    against the oracle, and against the reference where the rollouts agree."""
    fx = load(name)
    m = orc.Mlp.from_npz(weights_path("toy2"))
    N, H, k, K = int(fx["N"]), int(fx["H"]), int(fx["k"]), int(fx["K"])
    dt = np.float32(fx["dt"])
    eng1 = _engine(fx, N, 1)
    engH = _engine(fx, N, H)
    try:
        for it in range(int(fx["n_iter"])):
            pre = f"it{it}_"
            ref = fx[pre + "all_traj"]
            eng1.set_policy_samples(fx[pre + "mu_tmp"], fx[pre + "sigma_tmp"], fx[pre + "alpha_tmp"])
            for i in range(1, H + 1):
                eng1.propagate(np.ascontiguousarray(ref[:, i - 1, :]))
                r = eng1.get_rollouts()
                if i < H:
                    assert_close(ref[:, i - 1, :] + dt * r["qdot"], ref[:, i, :], 2e-4, f"next state {i}")
                assert_close(r["closest_dist_all"][:, 0], fx[pre + "closest_dist_all"][:, i - 1], 1e-5, f"distance {i}")
                assert_close(r["dot_products"][:, 0], fx[pre + "dot_products"][:, i - 1], 2e-4, f"dot {i}")
                assert_close(r["kernel_val_all"][:, 0], fx[pre + "kernel_val_all"][:, i - 1], 2e-4, f"phi*act {i}")
                assert_close(r["normal"][:, 0], fx[pre + "norm_basis_n"][:, i - 1], 2e-4, f"normal {i}")
                # against the oracle's modulation on the device's own distance/normal: 1e-5
                o = orc.propagate(m, ref[:, i - 1, :], fx["qf"], fx["obs"], N=N, H=1, dt=float(dt), k=k, ignored_links=[],
                                  mu_tmp=fx[pre + "mu_tmp"], sigma_tmp=fx[pre + "sigma_tmp"], alpha_tmp=fx[pre + "alpha_tmp"],
                                  prm=toy_prm(fx))
                assert_close(r["qdot"], o.qdot, 2e-4, f"qdot vs oracle {i}")
            engH.set_policy_samples(fx[pre + "mu_tmp"], fx[pre + "sigma_tmp"], fx[pre + "alpha_tmp"])
            engH.propagate(fx[pre + "q_cur"])
            r = engH.get_rollouts()
            assert_close(r["all_traj"], fx[pre + "all_traj"], 1e-2, "free-running rollouts")
            cost = engH.cost()
            ocost, _ = orc.evaluate_costs(r["all_traj"], r["closest_dist_all"], fx["qf"], np.zeros((3, 4), np.float32), fx["cost_q_min"],
                                          fx["cost_q_max"], terms=TOY_TERMS)
            assert_close(cost, ocost, 1e-5, "3-term cost vs oracle")
            mu, sg, al, mask, w = engH.weighted_update(float(fx["policy_upd_rate"]), float(fx["ker_thr"]), fx[pre + "mu_c"], fx[pre + "sigma_c"],
                                                       fx[pre + "alpha_c"], want_weights=True)
            omu, osg, oal, omask, ow = orc.shift_policy_means(cost, r["kernel_val_all"], None, fx[pre + "mu_c"], fx[pre + "sigma_c"],
                                                              fx[pre + "alpha_c"], fx[pre + "mu_tmp"], fx[pre + "sigma_tmp"], fx[pre + "alpha_tmp"],
                                                              float(fx["ker_thr"]), float(fx["policy_upd_rate"]), toy=True)
            assert np.array_equal(mask, omask)
            assert_close(w, ow, 2e-5, "weights vs oracle", floor=float(ow.max()))
            if K:
                assert_close(mu, omu, 2e-5, "mu_c vs oracle"); assert_close(sg, osg, 2e-5, "sigma_c"); assert_close(al, oal, 2e-5, "alpha_c")
    finally:
        eng1.close(); engH.close()


@pytest.mark.gpu
def test_gpu_toy_facade_like_the_driver():
    """scripts/standaloneToy2d.py:90-118 against the facade: same constructor, 4-tuple from propagate, the planner loop
    reaches the goal side of the arc without the best rollout ever being in collision."""
    import torch
    from optimalmodulationds_amd.mppi_toy import MPPI
    from optimalmodulationds_amd.robot_sdf import RobotSdfCollisionNet
    fx = load("toy2_arc_K0")
    nn_model = RobotSdfCollisionNet(in_channels=4, out_channels=1, layers=[256] * 4, skips=[])
    nn_model.load_weights(weights_path("toy2"))
    A = -1 * torch.diag(torch.ones(2))
    mppi = MPPI(torch.tensor([-5.0, 0.0]), torch.tensor([8.0, 0.0]), torch.zeros(4, 4), torch.from_numpy(fx["obs"]), 0.5, 10, 100, A, 0,
                nn_model, 1)
    mppi.Policy.sigma_c_nominal = 0.1; mppi.Policy.alpha_s = 0.75
    mppi.dst_thr = 0.25; mppi.ker_thr = 0.5; mppi.ignored_links = []
    mppi.Cost.q_min = -10 * torch.ones(2); mppi.Cost.q_max = 10 * torch.ones(2)
    mppi.Policy.set_samples(fx["it0_mu_tmp"], fx["it0_sigma_tmp"], fx["it0_alpha_tmp"])
    out = mppi.propagate()
    assert len(out) == 4
    assert_close(out[0].numpy(), fx["it0_all_traj"], 1e-2, "all_traj")
    assert_close(mppi.get_cost().numpy(), fx["it0_cost"], 1e-2, "cost")
    assert mppi.shift_policy_means() == 0
    for _ in range(60):
        mppi.Policy.sample_policy()
        mppi.propagate()
        mppi.get_cost()
        mppi.shift_policy_means()
        mppi.q_cur = mppi.q_cur + mppi.get_qdot('best') * 0.5
    assert float(mppi.q_cur[0]) > -5.0 and torch.isfinite(mppi.q_cur).all()

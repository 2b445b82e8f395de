"""Full-size parity, re-derived by the ORACLE (never by the device itself), on the three shapes the bench credits:

  * Franka shelf 4096 x 32, K = 10, the opt-in f16-screened step (omds_set_screening(ctx, 2, 0)) -- BASELINE configs[2]'s size, ReLU weights;
  * Franka shelf 1024 x 32, K = 10, the all-fp32 step (omds_set_screening(ctx, 0, 0)) -- bench.py's headline `value`;
  * Franka shelf 4096 x 32 with the 256x3 tanh network (tests/golden/weights/franka_tanh) -- configs[2] as BASELINE.json words it.

The device rollout is re-derived by the oracle at S random (rollout, step) states -- EVERY sampled row, nothing admitted -- and the
cost / MPPI weights / policy update are recomputed by the oracle from the device's own rollouts.  ReLU networks: the thresholded
distance must be the oracle's BITS (the device evaluates the network in the reference's arithmetic, oracle/chain_arith.c); the
modulated velocity meets north_star's plain 1e-5 on every row.  tanh: the device's tanhf is not numpy's, so the distance is held
to 1e-6 m and the velocity to the same plain bar.  `pytest -s` prints the table that profiles/rNN_parity_fullsize.txt keeps."""
import time

import numpy as np
import pytest

from helpers import RTOL, log_plain_bar, plain_bar, weights_path
from oracle import omds_oracle as orc

pytestmark = pytest.mark.gpu


def _err(a, b, scale=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    e = np.abs(a - b) / (scale if scale else 1.0)
    return float(e.max()), float(e.mean())


@pytest.mark.parametrize("kind,N,S,screen", [("franka", 4096, 1024, 2), ("franka", 1024, 1024, 0), ("franka_tanh", 4096, 512, 2)])
def test_fullsize_rollout_rederived_by_the_oracle(kind, N, S, screen):
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
    from optimalmodulationds_amd.engine import Engine
    H, k, K = 32, 5, 10
    m = orc.Mlp.from_npz(weights_path(kind))
    relu = m.act == "relu"
    obs, q0, qf, dh = scenes.shelf_scene(), scenes.FRANKA_Q0, scenes.FRANKA_QF, scenes.franka_dh_params()
    qmin, qmax = np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32)
    eng = Engine(7, N, H, k, max_obs=512)
    eng.set_mlp(m.W, m.b, act=m.act)
    eng.set_obstacles(obs)
    eng.params.dt, eng.params.dst_thr, eng.params.ignored_links = 0.5, 0.01, 0b111
    eng.push_params()
    eng.set_ds(qf)
    eng.set_cost(dh, qmin, qmax)
    eng.set_screening(screen, 0.0)         # 0: the all-fp32 step (the library's default, bench.py's `value`); 2: the opt-in screened step
    rng = np.random.RandomState(7)
    s = (np.arange(K) + 0.5) / K
    mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
    sg_c = np.ones(K, np.float32)
    al_c = rng.standard_normal((K, 7)).astype(np.float32)
    eng.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=99)
    mu, sg, al = eng.get_policy_samples()
    eng.propagate(q0)
    r = eng.get_rollouts()
    st_scr = eng.screen_stats()
    cost = eng.cost()
    nmu, nsg, nal, mask, w = eng.weighted_update(0.1, 0.1, mu_c, sg_c, al_c, want_weights=True)
    eng.close()
    if screen == 0:
        assert not st_scr["active"], st_scr
    else:
        assert st_scr["active"] and st_scr["fallbacks"] == 0, st_scr      # mode 2 screens at this size
    step_name = "all-fp32 step" if screen == 0 else "screened step"

    t0 = time.time()
    tt, hh = rng.randint(0, N, S), rng.randint(0, H, S)
    q = r["all_traj"][tt, hh]
    d, g, _, idx = orc.distance_repulsion_nn(m, q, obs, k, [0, 1, 2])
    st = orc.modulation_step(q, qf, d, g, mu[tt], sg[tt], al[tt], orc.Params(dst_thr=0.01))
    rows = []
    dist_dev, dist_orc = r["closest_dist_all"][tt, hh], (d - np.float32(0.01)).astype(np.float32)
    e_dist = _err(dist_dev, dist_orc)
    rows.append(("thresholded distance [m, absolute]", e_dist))
    same_bits = float(np.mean(dist_dev == dist_orc))
    e_norm = _err(r["normal"][tt, hh], st["ghat"])
    rows.append(("obstacle normal", e_norm))
    e_dot = _err(r["dot_products"][tt, hh], st["dot"])
    rows.append(("normal . nominal direction", e_dot))
    e_rbf = _err(r["kernel_val_all"][tt, hh], st["phi"])
    rows.append(("RBF kernel values", e_rbf))
    e_act = _err(r["kernel_activations"][tt, hh], st["act"])
    rows.append(("kernel activation", e_act))
    nxt = hh + 1 < H
    vel = (r["all_traj"][tt[nxt], hh[nxt] + 1] - q[nxt]) / np.float32(0.5)
    e_vel = _err(vel, st["u"][nxt])
    rows.append(("integrated velocity", e_vel))
    # the PLAIN north-star bar on EVERY sampled row with a next state: (q_next - q) / dt against the oracle's own step
    pb, e_rows = plain_bar(vel, st["u"][nxt])
    log_plain_bar(kind, f"full size {N} x {H}, {step_name}", "oracle", pb)
    ocost, _ = orc.evaluate_costs(r["all_traj"], r["closest_dist_all"], qf, dh, qmin, qmax)
    e_cost = _err(cost, ocost, scale=float(np.abs(ocost).max()))
    rows.append(("cost (all rollouts, rel. to max)", e_cost))
    omu, osg, oal, omask, ow = orc.shift_policy_means(cost, r["kernel_val_all"], r["kernel_activations"], mu_c, sg_c, al_c, mu, sg, al, 0.1, 0.1)
    e_w = _err(w, ow, scale=float(ow.max()))
    rows.append(("MPPI weights (rel. to max)", e_w))
    e_mean = _err(np.concatenate([nmu.ravel(), nsg, nal.ravel()]), np.concatenate([omu.ravel(), osg, oal.ravel()]))
    rows.append(("policy means after update", e_mean))
    scr = (f" (eps {st_scr['eps']:.4g}, {st_scr['candidates_per_rollout_step']:.2f} candidates and {st_scr['audit_rows_per_rollout_step']:.2f} audit rows per "
           f"rollout-step, audit max err {st_scr['audit_max_err']:.2e})") if screen != 0 else ""
    print(f"\nFranka shelf, {kind} weights, N={N}, H={H}, O={obs.shape[0]}, k={k}, K={K}, {step_name}{scr}; {S} sampled (rollout, step) states re-derived "
          f"by the oracle ({time.time() - t0:.1f} s), every one of them compared")
    print("quantity                             max err      mean err")
    for name, (mx, mean) in rows:
        print(f"{name:36s} {mx:.3e}    {mean:.3e}")
    print(f"thresholded distance: {100.0 * same_bits:.2f} % of the sampled states carry the oracle's bits")
    print(f"plain 1e-5 bar (integrated velocity vs the oracle, all {pb['rows']} sampled rows with a next state, nothing admitted): "
          f"{pb['plain']} rows = {100.0 * pb['plain'] / pb['rows']:.2f} %, worst row {pb['worst']:.2e} (of which (q_next - q) / dt loses ~1e-6)")
    print("update mask identical:", bool(np.array_equal(mask, omask)), " finite outputs:", bool(np.isfinite(r["all_traj"]).all()))
    assert np.isfinite(r["all_traj"]).all() and np.array_equal(mask, omask)
    if relu:
        assert same_bits == 1.0, (same_bits, e_dist)
    assert e_dist[0] <= 1e-6, e_dist
    assert pb["plain"] == pb["rows"], pb
    assert e_norm[0] <= 2e-5 and e_dot[0] <= 2e-5 and e_rbf[0] <= 1e-5, (e_norm, e_dot, e_rbf)
    assert e_vel[0] <= 1e-5 * max(1.0, float(np.abs(st["u"]).max())) + 4e-6 * max(1.0, float(np.abs(q).max())) / 0.5, e_vel   # (q_next - q) / dt loses bits
    assert e_cost[0] <= 1e-6 and e_w[0] <= 1e-6 and e_mean[0] <= 2e-6, (e_cost, e_w, e_mean)

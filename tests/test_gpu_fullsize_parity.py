"""Full-size parity at BASELINE configs[2]'s shape (Franka shelf, 4096 rollouts x 32 horizon, K = 10, the default -- screened --
step): the device rollout is re-derived by the oracle at 1024 random (rollout, step) states, and the cost / MPPI weights /
policy update are recomputed by the oracle from the device's own rollouts.  The bars are the maxima this comparison has shown
since round 1, with headroom of 2-4x: distance 1e-6 (absolute, metres), obstacle normal and integrated velocity 2e-5 on rows
without a ReLU pre-activation within 5e-6 of zero; such rows must match the oracle under SOME admissible assignment of the
ambiguous masks at the same 2e-5.  `pytest -s` prints the table that profiles/rNN_parity_fullsize.txt keeps."""
import time

import numpy as np
import pytest

from helpers import DIST_ULP, RTOL, log_plain_bar, plain_bar, velocity_envelope, weights_path
from oracle import omds_oracle as orc

pytestmark = pytest.mark.gpu


def _err(a, b, sel=None, scale=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if sel is not None:
        a, b = a[sel], b[sel]
    e = np.abs(a - b) / (scale if scale else 1.0)
    return float(e.max()), float(e.mean())


@pytest.mark.parametrize("N,S", [(4096, 1024)])
def test_fullsize_rollout_rederived_by_the_oracle(N, S):
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
    from optimalmodulationds_amd.engine import Engine
    H, k, K = 32, 5, 10
    m = orc.Mlp.from_npz(weights_path("franka"))
    obs, q0, qf, dh = scenes.shelf_scene(), scenes.FRANKA_Q0, scenes.FRANKA_QF, scenes.franka_dh_params()
    qmin, qmax = np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32)
    eng = Engine(7, N, H, k, max_obs=512)
    eng.set_mlp(m.W, m.b)
    eng.set_obstacles(obs)
    eng.params.dt, eng.params.dst_thr, eng.params.ignored_links = 0.5, 0.01, 0b111
    eng.push_params()
    eng.set_ds(qf)
    eng.set_cost(dh, qmin, qmax)
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
    assert st_scr["active"] and st_scr["fallbacks"] == 0, st_scr      # the default path at this size is the screened step

    t0 = time.time()
    tt, hh = rng.randint(0, N, S), rng.randint(0, H, S)
    q = r["all_traj"][tt, hh]
    d, g, _, idx = orc.distance_repulsion_nn(m, q, obs, k, [0, 1, 2])
    st = orc.modulation_step(q, qf, d, g, mu[tt], sg[tt], al[tt], orc.Params(dst_thr=0.01))
    ok = orc.rollout_relu_margin(m, q, obs, idx) >= 5e-6
    rows = []
    e_dist = _err(r["closest_dist_all"][tt, hh], d - np.float32(0.01))
    rows.append(("thresholded distance [m, absolute]", e_dist))
    e_norm = _err(r["normal"][tt, hh], st["ghat"], ok)
    rows.append(("obstacle normal (unflagged)", e_norm))
    gn = r["normal"][tt, hh]
    worst_alt, n_unmatched = 0.0, 0
    for i in np.nonzero(~ok)[0]:
        alts = orc.blended_gradient_alternatives(m, q[i], obs, idx[i], 5e-6)
        alts = alts / np.linalg.norm(alts, axis=1, keepdims=True)
        e_best = float(np.abs(alts - gn[i]).max(axis=1).min())
        worst_alt = max(worst_alt, e_best)
        n_unmatched += e_best > 2e-5
    e_dot = _err(r["dot_products"][tt, hh], st["dot"], ok)
    rows.append(("normal . nominal direction (unfl.)", e_dot))
    e_rbf = _err(r["kernel_val_all"][tt, hh], st["phi"])
    rows.append(("RBF kernel values", e_rbf))
    e_act = _err(r["kernel_activations"][tt, hh], st["act"], ok)
    rows.append(("kernel activation (unflagged)", e_act))
    nxt = hh + 1 < H
    vel = (r["all_traj"][tt[nxt], hh[nxt] + 1] - q[nxt]) / np.float32(0.5)
    e_vel = _err(vel, st["u"][nxt], ok[nxt])
    rows.append(("integrated velocity (unflagged)", e_vel))
    # the PLAIN north-star bar on EVERY sampled row with a next state (flagged rows included, no envelope, no mask alternatives): the
    # integrated velocity (q_next - q) / dt against the oracle's own step; then what the others need (helpers.plain_bar)
    prm = orc.Params(dst_thr=0.01)
    sel = np.nonzero(nxt)[0]
    lo, hi = velocity_envelope(q[sel], qf, d[sel], (g[sel], gn[sel]), mu[tt][sel], sg[tt][sel], al[tt][sel], prm, DIST_ULP * max(1.0, float(np.abs(d).max())))
    pad = RTOL * max(1.0, float(np.abs(hi).max())) + 4e-6 * max(1.0, float(np.abs(q).max())) / 0.5      # (q_next - q) / dt loses bits
    in_env = ((vel >= lo - pad) & (vel <= hi + pad)).all(axis=1) & ok[sel]
    pb, e_rows = plain_bar(vel, st["u"][sel], in_env)
    log_plain_bar("franka", f"full size {N} x {H}, screened step", "oracle", pb)
    ocost, _ = orc.evaluate_costs(r["all_traj"], r["closest_dist_all"], qf, dh, qmin, qmax)
    e_cost = _err(cost, ocost, scale=float(np.abs(ocost).max()))
    rows.append(("cost (all rollouts, rel. to max)", e_cost))
    omu, osg, oal, omask, ow = orc.shift_policy_means(cost, r["kernel_val_all"], r["kernel_activations"], mu_c, sg_c, al_c, mu, sg, al, 0.1, 0.1)
    e_w = _err(w, ow, scale=float(ow.max()))
    rows.append(("MPPI weights (rel. to max)", e_w))
    e_mean = _err(np.concatenate([nmu.ravel(), nsg, nal.ravel()]), np.concatenate([omu.ravel(), osg, oal.ravel()]))
    rows.append(("policy means after update", e_mean))
    print(f"\nFranka shelf, N={N}, H={H}, O={obs.shape[0]}, k={k}, K={K}, screened step (eps {st_scr['eps']:.4g}, "
          f"{st_scr['candidates_per_rollout_step']:.2f} candidates and {st_scr['audit_rows_per_rollout_step']:.2f} audit rows per rollout-step, "
          f"audit max err {st_scr['audit_max_err']:.2e}); {S} sampled (rollout, step) states re-derived by the oracle ({time.time() - t0:.1f} s); "
          f"{100 * (1 - ok.mean()):.2f} % of them have a ReLU pre-activation within 5e-6 of zero")
    print("quantity                             max err      mean err")
    for name, (mx, mean) in rows:
        print(f"{name:36s} {mx:.3e}    {mean:.3e}")
    print(f"obstacle normal, flagged rows under the best admissible mask assignment: max err {worst_alt:.3e}, rows above 2e-5: {n_unmatched} of {int((~ok).sum())}")
    print(f"plain 1e-5 bar (integrated velocity vs the oracle, all {pb['rows']} sampled rows with a next state, no envelope, no mask alternatives): "
          f"{pb['plain']} rows = {100.0 * pb['plain'] / pb['rows']:.2f} % (worst of them {pb['worst_plain']:.2e}); {pb['envelope']} more inside the "
          f"+-{DIST_ULP:.0e} distance envelope; {pb['mask']} need another admissible ReLU-mask assignment (worst row {pb['worst']:.2e})")
    print("update mask identical:", bool(np.array_equal(mask, omask)), " finite outputs:", bool(np.isfinite(r["all_traj"]).all()))
    assert pb["plain"] >= 0.95 * pb["rows"], pb
    assert np.isfinite(r["all_traj"]).all() and np.array_equal(mask, omask)
    assert e_dist[0] <= 1e-6, e_dist
    assert e_norm[0] <= 2e-5 and n_unmatched == 0, (e_norm, worst_alt, n_unmatched)
    assert e_dot[0] <= 2e-5 and e_rbf[0] <= 1e-5, (e_dot, e_rbf)
    assert e_vel[0] <= 2e-5, e_vel
    assert e_cost[0] <= 1e-6 and e_w[0] <= 1e-6 and e_mean[0] <= 2e-6, (e_cost, e_w, e_mean)

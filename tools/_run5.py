import sys, time, numpy as np
sys.path.insert(0, '.')
import bench, argparse
from optimalmodulationds_amd.engine import Engine
w, W, b, obs, q0, qf, dh, qmin, qmax = bench.setup("franka_shelf_1024x32", 0)
N, H, n, K = w["N"], w["H"], 7, 10
eng = Engine(n, N, H, w["k"], max_obs=max(64, obs.shape[0]))
eng.set_mlp(W, b); eng.set_obstacles(obs)
p = eng.params; p.dt, p.dst_thr = w["dt"], w["dst_thr"]; p.ignored_links = 7; eng.push_params()
eng.set_ds(qf); eng.set_cost(dh, qmin, qmax)
rng = np.random.RandomState(1234)
s = (np.arange(K) + 0.5) / K
mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, n))).astype(np.float32)
sg_c = np.full(K, w["sigma"], np.float32); al_c = rng.standard_normal((K, n)).astype(np.float32)
q_cur = q0.copy()
ts = []
for it in range(230):
    t0 = time.perf_counter()
    eng.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, w["alpha_s"], K, seed=1234 * 1000003 + it, rollout_offset=0)
    eng.propagate(q_cur); eng.cost(fetch=False)
    mu_c, sg_c, al_c, mask, qd_w, _, _ = eng.weighted_update_sharded(0.1, w["ker_thr"], mu_c, sg_c, al_c)
    q_cur = (q_cur + 0.1 * w["dt"] * qd_w).astype(np.float32)
    ts.append((time.perf_counter() - t0) * 1e3)
    st = eng.screen_stats()
    if ts[-1] > 7.5: print(it, round(ts[-1], 2), st["fallbacks"], round(st["eps"], 5), round(st["max_err_seen"], 5), round(st["audit_max_err"], 5))
print("median", np.median(ts), "fallbacks", eng.screen_stats()["fallbacks"], eng.screen_stats())
print([round(x, 2) for x in ts[::10]])

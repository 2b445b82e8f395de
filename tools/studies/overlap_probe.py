"""Feasibility probe: G independent contexts (own HIP stream each) of N/G rollouts propagating concurrently from G host
threads, against one context of N rollouts -- does overlapping one group's power-bound k_screen with the other groups'
latency-bound k_exact / k_tail_sel pay?  usage: OMDS_SCREEN_CUS=160 python tests/overlap_probe.py [G]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from test_gpu_screen import _engine

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, H, K, iters = 1024, 32, 10, 12

def setup(n):
    e, m, obs, q0, qf = _engine(n, H)
    rng = np.random.RandomState(5)
    s = (np.arange(K) + 0.5) / K
    mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
    e.sample_policy(mu_c, np.ones(K, np.float32), rng.standard_normal((K, 7)).astype(np.float32), 0.0, 0.0, 3.0, K, seed=1)
    e.set_screening(1)
    e.propagate(q0); e.propagate(q0)
    return e, q0

def run(e, q0, n_it):
    for _ in range(n_it):
        e.propagate(q0)

e1, q0 = setup(N)
t = time.time(); run(e1, q0, iters); t1 = (time.time() - t) / iters
print(f"one context, N={N}: {t1 * 1e3:.3f} ms per propagate = {t1 / H * 1e6:.1f} us per step", e1.screen_stats())
e1.close()
es = [setup(N // G) for _ in range(G)]
ths = [threading.Thread(target=run, args=(e, q, iters)) for e, q in es]
t = time.time()
for th in ths: th.start()
for th in ths: th.join()
tg = (time.time() - t) / iters
print(f"{G} contexts x N={N // G} concurrently: {tg * 1e3:.3f} ms per propagate of all = {tg / H * 1e6:.1f} us per step  (x{t1 / tg:.2f})")

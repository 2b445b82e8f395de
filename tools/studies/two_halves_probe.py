"""Feasibility probe: does the all-fp32 propagate of 1024 rollouts run faster as TWO independent halves on two streams?
Two Engine contexts of 512 rollouts driven by two host threads (ctypes releases the GIL inside omds_propagate) against one context of
1024: while one half sits in its latency-bound k_tail the other half's k_pass1 has the GPU.  usage: python tools/studies/two_halves_probe.py"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from optimalmodulationds_amd import scenes   # noqa: E402
from optimalmodulationds_amd.engine import Engine   # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "weights", "franka.npz"))
W = [z[f"W{i}"] for i in range(5)]; b = [z[f"b{i}"] for i in range(5)]
H, k, K = 32, 5, 3
obs = scenes.shelf_scene()


def make(N, seed):
    e = Engine(7, N, H, k, max_obs=512)
    e.set_mlp(W, b); e.set_obstacles(obs)
    e.params.dt = 0.01; e.params.dst_thr = 0.5
    e.push_params(); e.set_ds(scenes.FRANKA_QF)
    rng = np.random.RandomState(seed)
    mu = (scenes.FRANKA_Q0 + 0.2 * rng.standard_normal((N, K, 7))).astype(np.float32)
    e.set_policy_samples(mu, np.ones((N, K), np.float32), rng.standard_normal((N, K, 7)).astype(np.float32))
    q0 = (scenes.FRANKA_Q0 + 0.3 * rng.standard_normal((N, 7))).astype(np.float32)
    return e, q0


def run(engs, iters):
    def work(e, q):
        for _ in range(iters):
            e.propagate(q)
    ths = [threading.Thread(target=work, args=eq) for eq in engs]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    return (time.perf_counter() - t0) / iters * 1e3


one = [make(1024, 0)]
two = [make(512, 1), make(512, 2)]
four = [make(256, 3 + i) for i in range(4)]
for name, engs in (("1 x 1024", one), ("2 x 512", two), ("4 x 256", four)):
    run(engs, 2)
for rnd in range(3):
    for name, engs in (("1 x 1024", one), ("2 x 512", two), ("4 x 256", four)):
        ms = run(engs, 6)
        print(f"{name}: {ms:.3f} ms per propagate of 1024 rollouts x {H} steps ({1024 * H / ms * 1e3 / 1e6:.3f} M rollout-steps/s)")

"""Candidates per rollout and step of the screened step as a function of the screening bound eps (GPU; prints the screening
counters after one propagate per setting).  usage: python tests/eps_sweep.py"""
import os
import sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from test_gpu_screen import _engine
for N in (1024, 4096):
    for eps in (1.5e-2, 1.0e-2, 7.5e-3, 6e-3, 5e-3, 4e-3):
        e, m, obs, q0, qf = _engine(N, 32)
        K = 10
        rng = np.random.RandomState(5)
        s = (np.arange(K) + 0.5) / K
        mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
        e.set_screening(1, eps)
        e.sample_policy(mu_c, np.ones(K, np.float32), rng.standard_normal((K, 7)).astype(np.float32), 0.0, 0.0, 3.0, K, seed=100)
        e.propagate(q0)
        print(N, eps, e.screen_stats())
        e.close()

"""k_exact launch time as a function of the candidate count (GPU, under rocprofv3 --kernel-trace --stats):
usage: EPS=0.0075 rocprofv3 ... -- python3 tests/exact_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from test_gpu_screen import _engine  # noqa: E402

eps = float(os.environ.get("EPS", "0.015"))
e, m, obs, q0, qf = _engine(1024, 16)
rng = np.random.RandomState(5)
K = 10
s = (np.arange(K) + 0.5) / K
mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
e.set_screening(1, eps)
for it in range(3):
    e.sample_policy(mu_c, np.ones(K, np.float32), rng.standard_normal((K, 7)).astype(np.float32), 0.0, 0.0, 3.0, K, seed=100 + it)
    e.set_screening(1, eps)
    e.propagate(q0)
print(eps, e.screen_stats())

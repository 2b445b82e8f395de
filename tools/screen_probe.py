#!/usr/bin/env python3
"""Timing probe for the fp16 screening kernel: k_screen<NHH> on random ReLU networks of 2..5 hidden layers over the same
N x O pairs, to separate the fixed per-tile cost from the per-layer cost.  Run under
    rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/screen_probe.py
and read the per-kernel averages (tools/rocprof_summary.py stats)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimalmodulationds_amd import scenes  # noqa: E402
from optimalmodulationds_amd.engine import Engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
obs = scenes.shelf_scene()
rng = np.random.RandomState(0)
q = (scenes.FRANKA_Q0 + 0.3 * rng.standard_normal((N, 7))).astype(np.float32)
for hidden in (2, 3, 4, 5):
    dims = [30] + [256] * hidden + [9]
    W = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    b = [0.1 * rng.standard_normal(dims[i + 1]).astype(np.float32) for i in range(len(dims) - 1)]
    e = Engine(7, N, 2, 5, max_obs=512)
    e.set_mlp(W, b)
    e.set_obstacles(obs)
    for _ in range(6):
        e.screen_mindist(q)
    e.close()
print("done")

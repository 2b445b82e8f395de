#!/usr/bin/env python3
"""Diagnostic (GPU box): error of the fp16 screening values against the fp32 pass-1 matrix as a function of the distance
itself, on states like the ones rollouts visit and on uniformly drawn joint states.
usage: python tools/screen_debug.py [n_states]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optimalmodulationds_amd import scenes  # noqa: E402
from optimalmodulationds_amd.engine import Engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
z = np.load(os.path.join(sys.path[0], "tests/golden/weights/franka.npz"))
nl = len([k for k in z.files if k.startswith("W")])
e = Engine(7, N, 2, 5, max_obs=512)
e.set_mlp([z[f"W{i}"] for i in range(nl)], [z[f"b{i}"] for i in range(nl)])
e.set_obstacles(scenes.shelf_scene())
rng = np.random.RandomState(0)
q0, qf = scenes.FRANKA_Q0, scenes.FRANKA_QF
sets = {"q0..qf +- 0.3": (q0 + (qf - q0) * rng.rand(N, 1) + 0.3 * rng.standard_normal((N, 7))).astype(np.float32),
        "q0..1.5 qf +- 0.6": (q0 + 1.5 * (qf - q0) * rng.rand(N, 1) + 0.6 * rng.standard_normal((N, 7))).astype(np.float32),
        "uniform +-2.9": rng.uniform(-2.9, 2.9, (N, 7)).astype(np.float32)}
edges = [-1.0, 0.0, 0.1, 0.2, 0.4, 0.7, 1.0, 1.5, 3.0]
for name, q in sets.items():
    _, _, ref, _ = e.dist_grad(q, want_mindist=True)
    apx = e.screen_mindist(q)
    err = np.abs(apx - ref)
    print(f"{name}: max err {err.max():.2e}, mean {err.mean():.2e}; by fp32 distance bin:")
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = (ref >= lo) & (ref < hi)
        if sel.any():
            print(f"   d in [{lo:4.1f},{hi:4.1f}): n {int(sel.sum()):8d}  max err {err[sel].max():.2e}  max err/(0.1+|d|) {(err[sel] / (0.1 + np.abs(ref[sel]))).max():.2e}")

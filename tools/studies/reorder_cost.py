"""What a calibration costs on the host clock: the first propagates of a fresh context (calibration + unit re-sort inside the first,
the refinement of the order behind it, then steady state), Franka shelf 1024 x 32.  python tools/studies/reorder_cost.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from optimalmodulationds_amd import scenes  # noqa: E402
from optimalmodulationds_amd.engine import Engine  # noqa: E402

z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "weights", "franka.npz"))
n = len([k for k in z.files if k.startswith("W")])
W, b = [z[f"W{i}"] for i in range(n)], [z[f"b{i}"] for i in range(n)]
for mode, name in ((1, "screened"), (0, "fp32 only")):
    e = Engine(7, 1024, 32, 5, max_obs=320)
    e.set_mlp(W, b)
    e.set_obstacles(scenes.shelf_scene())
    e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.5, 0.01, 0b111
    e.push_params()
    e.set_ds(scenes.FRANKA_QF)
    e.set_screening(mode)
    ts = []
    for it in range(6):
        e.sample_policy(None, None, None, 0, 0, 0, 0, seed=it)
        t0 = time.perf_counter()
        e.propagate(np.asarray(scenes.FRANKA_Q0, np.float32))
        ts.append((time.perf_counter() - t0) * 1e3)
    st = e.screen_stats()
    print(f"{name}: propagate ms {[round(t, 2) for t in ts]}  reorders {st['unit_reorders']}  never fired {st['units_never_fired']}")
    e.close()

"""Soak run of the facade planner loop (examples/franka_planner_loop.py) on the GPU: many iterations with moving
obstacles, kernel adding and repeated context creation, watching device memory and finiteness.
usage: python tools/soak.py [iterations] [contexts]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import franka_planner_loop as ex  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ctxs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
free0, total = torch.cuda.mem_get_info()
t0 = time.time()
for c in range(ctxs):
    mppi = ex.main(iters=iters // ctxs, n_traj=(1024, 256, 40, 2048)[c % 4], horizon=(32, 16, 10, 8)[c % 4], moving=True, quiet=True)
    assert torch.isfinite(mppi.q_cur).all() and torch.isfinite(mppi.all_traj).all()
    assert np.isfinite(mppi.Policy.mu_c.numpy()).all() and np.isfinite(mppi.Policy.alpha_c.numpy()).all()
    mppi._engine.close()
    mppi.nn_model._engine and mppi.nn_model._engine.close()
    del mppi
    free1, _ = torch.cuda.mem_get_info()
    print(f"context {c}: device memory in use by this process changed by {(free0 - free1) / 2**20:+.1f} MiB since start")
print(f"soak ok: {iters} iterations in {time.time() - t0:.1f} s")

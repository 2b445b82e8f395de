"""Propagate latency of small (latency-bound) shapes: reference defaults and the integrator tick."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimalmodulationds_amd import scenes
from optimalmodulationds_amd.engine import Engine

def run(kind, n, N, H, k, obs, q0, qf, K=4):
    z = np.load(os.path.join(ROOT, "tests", "golden", "weights", kind + ".npz"))
    nl = len([x for x in z.files if x.startswith("W")])
    eng = Engine(n, N, H, k, max_obs=max(8, len(obs)))
    eng.set_mlp([z[f"W{i}"] for i in range(nl)], [z[f"b{i}"] for i in range(nl)])
    eng.set_obstacles(obs); eng.params.dt = 0.3; eng.push_params(); eng.set_ds(qf)
    rng = np.random.RandomState(0)
    eng.sample_policy(rng.standard_normal((K, n)).astype(np.float32), np.ones(K, np.float32),
                      rng.standard_normal((K, n)).astype(np.float32), 0, 0, 1.0, K, seed=1)
    for _ in range(3): eng.propagate(q0)
    t = time.perf_counter()
    for _ in range(20): eng.propagate(q0)
    dt = (time.perf_counter() - t) / 20
    eng.close()
    return dt * 1e3

shelf = scenes.shelf_scene()
cases = [("planar2 C1 64x16 O=1", "planar2", 2, 64, 16, 1, scenes.planar2_scene(1), np.array([-3.14, 0], np.float32), np.array([3.14, 0], np.float32)),
         ("franka integrator 1x2 shelf", "franka", 7, 1, 2, 5, shelf, scenes.FRANKA_Q0, scenes.FRANKA_QF),
         ("franka planner 40x10 shelf", "franka", 7, 40, 10, 5, shelf, scenes.FRANKA_Q0, scenes.FRANKA_QF),
         ("franka 256x8 shelf", "franka", 7, 256, 8, 5, shelf, scenes.FRANKA_Q0, scenes.FRANKA_QF)]
for name, kind, n, N, H, k, obs, q0, qf in cases:
    print(f"{name:32s} persistent={os.environ.get('OMDS_PERSISTENT','0')}  propagate {run(kind, n, N, H, k, obs, q0, qf):8.3f} ms")

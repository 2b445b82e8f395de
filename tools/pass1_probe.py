"""Times k_pass1 alone for a given number of rollouts (dist_grad batches) -- used with rocprofv3."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimalmodulationds_amd import scenes
from optimalmodulationds_amd.engine import Engine
z = np.load(os.path.join(ROOT, "tests", "golden", "weights", "franka.npz"))
W = [z[f"W{i}"] for i in range(5)]; b = [z[f"b{i}"] for i in range(5)]
for B in [int(x) for x in sys.argv[1:]]:
    eng = Engine(7, B, 1, 5, max_obs=512)
    eng.set_mlp(W, b); eng.set_obstacles(scenes.shelf_scene())
    q = (scenes.FRANKA_Q0 + 0.3 * np.random.RandomState(0).standard_normal((B, 7))).astype(np.float32)
    for _ in range(6):
        eng.dist_grad(q)
    eng.close()

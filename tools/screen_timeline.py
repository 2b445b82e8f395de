import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from optimalmodulationds_amd import scenes
from optimalmodulationds_amd.engine import Engine
z = np.load(os.path.join(sys.path[0], "tests/golden/weights/franka.npz"))
nl = len([k for k in z.files if k.startswith("W")])
class m: W = [z[f"W{i}"] for i in range(nl)]; b = [z[f"b{i}"] for i in range(nl)]
N = 1024
e = Engine(7, N, 2, 5, max_obs=512)
e.set_mlp(m.W, m.b); e.set_obstacles(scenes.shelf_scene())
rng = np.random.RandomState(0)
q = (scenes.FRANKA_Q0 + 0.3 * rng.standard_normal((N, 7))).astype(np.float32)
for _ in range(4): e.screen_mindist(q)

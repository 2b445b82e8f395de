import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from optimalmodulationds_amd import scenes
from optimalmodulationds_amd.engine import Engine
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
z = np.load(os.path.join(ROOT, "tests", "golden", "weights", "franka.npz"))
W = [z[f"W{i}"] for i in range(5)]; b = [z[f"b{i}"] for i in range(5)]
eng = Engine(7, 1024, 32, 5, max_obs=512)
eng.set_mlp(W, b); eng.set_obstacles(scenes.shelf_scene()); eng.set_ds(scenes.FRANKA_QF)
eng.params.dt = 0.5; eng.params.dst_thr = 0.01; eng.params.ignored_links = 7; eng.push_params()
K = 10
rng = np.random.RandomState(0)
mu = rng.standard_normal((K, 7)).astype(np.float32); sg = np.ones(K, np.float32); al = rng.standard_normal((K, 7)).astype(np.float32)
if len(sys.argv) > 1 and sys.argv[1] == "prof":
    eng.prof_enable(True)
for it in range(3):
    eng.sample_policy(mu, sg, al, 0.0, 0.0, 3.0, K, seed=it)
    eng.propagate(scenes.FRANKA_Q0)
eng.close()

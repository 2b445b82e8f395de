import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from optimalmodulationds_amd import scenes
from optimalmodulationds_amd.engine import Engine
from oracle import omds_oracle as orc
m = orc.Mlp.from_npz(os.path.join(sys.path[0], "tests/golden/weights/franka.npz"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
e = Engine(7, N, 2, 5, max_obs=512)
e.set_mlp(m.W, m.b); obs = scenes.shelf_scene(); e.set_obstacles(obs)
rng = np.random.RandomState(0)
q = (scenes.FRANKA_Q0 + 0.3 * rng.standard_normal((N, 7))).astype(np.float32)
_, _, ref, _ = e.dist_grad(q, want_mindist=True)
apx = e.screen_mindist(q)
err = np.abs(apx - ref).reshape(-1)
print("N", N, "rows", err.size, "max err", err.max(), "mean", err.mean(), "frac>5e-3", (err > 5e-3).mean())
bad = np.where(err > 5e-3)[0]
print("first bad rows", bad[:20], "tile", bad[:20] // 256, "wave", (bad[:20] % 256) // 32)
# per tile error
nt = (err.size + 255) // 256
for t in range(min(nt, 12)):
    seg = err[t * 256:(t + 1) * 256]
    print("tile", t, "max", seg.max(), "bad", (seg > 5e-3).sum(), " per-wave bad", [(seg[w*32:(w+1)*32] > 5e-3).sum() for w in range(8)])
print(apx.reshape(-1)[:8], ref.reshape(-1)[:8])
e.set_screening(1)
e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.5, 0.01, 0b111
e.push_params(); e.set_ds(scenes.FRANKA_QF)
e.sample_policy(None, None, None, 0, 0, 0, 0, seed=1)
e.propagate(scenes.FRANKA_Q0)
print(e.screen_stats())
for lim in (1.0, 2.0, 3.14159):
    q2 = rng.uniform(-lim, lim, (N, 7)).astype(np.float32)
    _, _, ref2, _ = e.dist_grad(q2, want_mindist=True)
    apx2 = e.screen_mindist(q2)
    er = np.abs(apx2 - ref2)
    i = np.unravel_index(np.argmax(er), er.shape)
    print("uniform +-%.2f: max err %.4g mean %.3g  at %s ref %.4g apx %.4g  |ref| max %.3g" % (lim, er.max(), er.mean(), i, ref2[i], apx2[i], np.abs(ref2[ref2 < 1e5]).max()))

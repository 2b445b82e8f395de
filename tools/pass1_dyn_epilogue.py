"""Per-WAVE stamps around one level's product and epilogue of the compacted k_pass1 (diagnostic build with -DOMDS_TIMELINE_E=<level>:
    make -C optimalmodulationds_amd/csrc timeline_e [TLE_LEVEL=1]   ->  libomds_hip_tle.so).
Every wave's lane 0 records wall_clock64 (100 MHz) at: 0 product start, 1 product done, 2 bias / ReLU / ballot done (in front of barrier 1),
3 barrier 1 passed, 4 level stored (in front of barrier 2), 5 barrier 2 passed.  Prints where the epilogue's time goes: a wave's own
instructions, or waiting at the barriers for the workgroup's slowest wave.   usage: python tools/pass1_dyn_epilogue.py [rollouts]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optimalmodulationds_amd._lib as L   # noqa: E402
L.LIB_PATH = os.path.join(ROOT, "optimalmodulationds_amd", "csrc", "libomds_hip_tle.so")
from optimalmodulationds_amd import scenes   # noqa: E402
from optimalmodulationds_amd.engine import Engine   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
z = np.load(os.path.join(ROOT, "tests", "golden", "weights", "franka.npz"))
W = [z[f"W{i}"] for i in range(5)]; b = [z[f"b{i}"] for i in range(5)]
eng = Engine(7, B, 1, 5, max_obs=512)
eng.set_mlp(W, b); obs = scenes.shelf_scene(); eng.set_obstacles(obs)
rng = np.random.RandomState(0)
q0, qf = np.asarray(scenes.FRANKA_Q0, np.float32), np.asarray(scenes.FRANKA_QF, np.float32)
q = (q0 + rng.rand(B, 1).astype(np.float32) * (qf - q0) + 0.3 * rng.standard_normal((B, 7))).astype(np.float32)
for _ in range(4):
    eng.dist_grad(q)
total = B * obs.shape[0]
tiles64 = total // 64
n_big = tiles64 - 256 if total >= 64 * 1024 else 0
nwg = n_big + (total - n_big * 64 + 31) // 32
buf = np.zeros((nwg * 4, 16), dtype=np.uint64)
fn = eng.lib.omds_timeline_fetch
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert fn(eng.h, buf.ctypes.data, nwg * 4) == 0
eng.close()
t = buf.reshape(nwg, 8, 8)[:n_big].astype(np.int64) / 100.0   # [workgroup][wave][stamp], us; 64-row tiles only
ok = (t[:, :, :6] > 0).all(axis=(1, 2))
t = t[ok]
print(f"{ok.sum()} workgroups of 64 rows with complete stamps")
st = t[:, :, 0].min(axis=1, keepdims=True)
def show(name, v):   # medians: a few workgroups carry a stamp of an earlier launch in a slot
    print(f"   {name:58s} median {np.median(v):6.2f}  p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f} us")
show("product, per wave (start -> done)", (t[:, :, 1] - t[:, :, 0]).ravel())
show("product, skew inside the workgroup (last - first wave done)", t[:, :, 1].max(axis=1) - t[:, :, 1].min(axis=1))
show("bias / ReLU / ballot, per wave", (t[:, :, 2] - t[:, :, 1]).ravel())
show("wait at barrier 1, per wave", (t[:, :, 3] - t[:, :, 2]).ravel())
show("barrier 1: last arrival -> first release", t[:, :, 3].min(axis=1) - t[:, :, 2].max(axis=1))
show("rank + level stores, per wave", (t[:, :, 4] - t[:, :, 3]).ravel())
show("wait at barrier 2, per wave", (t[:, :, 5] - t[:, :, 4]).ravel())
show("epilogue, workgroup (first wave's product done -> barrier 2)", t[:, :, 5].max(axis=1) - t[:, :, 1].min(axis=1))
show("epilogue, workgroup (LAST wave's product done -> barrier 2)", t[:, :, 5].max(axis=1) - t[:, :, 1].max(axis=1))

"""Per-workgroup phase timeline of k_exact's 32-row LIST tile (diagnostic build: make -C optimalmodulationds_amd/csrc timeline).
One screened propagate with a bound small enough for one tile per workgroup; stamps as in tools/pass1_timeline.py.
usage: python tools/exact_timeline.py [eps]"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optimalmodulationds_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "optimalmodulationds_amd", "csrc", "libomds_hip_tl.so")
from optimalmodulationds_amd import scenes
from optimalmodulationds_amd.engine import Engine

eps = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0075
N = 1024
z = np.load(os.path.join(ROOT, "tests", "golden", "weights", "franka.npz"))
W = [z[f"W{i}"] for i in range(5)]; b = [z[f"b{i}"] for i in range(5)]
eng = Engine(7, N, 1, 5, max_obs=512)
eng.set_mlp(W, b); eng.set_obstacles(scenes.shelf_scene())
eng.params.dt, eng.params.dst_thr, eng.params.ignored_links = 0.5, 0.01, 0b111
eng.push_params(); eng.set_ds(scenes.FRANKA_QF)
eng.sample_policy(None, None, None, 0, 0, 0, 0, seed=1)
q = (scenes.FRANKA_Q0 + 0.3 * np.random.RandomState(0).standard_normal((N, 7))).astype(np.float32)
for _ in range(3):
    eng.set_screening(1, eps)
    eng.propagate(q)
print(eng.screen_stats())
nwg = 256
buf = np.zeros((nwg, 16), dtype=np.uint64)
fn = eng.lib.omds_timeline_fetch
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert fn(eng.h, buf.ctypes.data, nwg) == 0
eng.close()
t = buf[:, :12].astype(np.int64)
u = (t - t[:, 0].min()) / 100.0
ph = {"entry (after launch start)": u[:, 0], "L1: loads back": u[:, 8] - u[:, 0], "L1: LDS written": u[:, 9] - u[:, 8], "L1: barrier": u[:, 1] - u[:, 9],
      "GEMM 1": u[:, 6] - u[:, 1], "epilogue 1": u[:, 2] - u[:, 6], "layer 2": u[:, 3] - u[:, 2], "layer 3": u[:, 4] - u[:, 3],
      "last: MFMA loop": u[:, 10] - u[:, 4], "last: rest + mask flush": u[:, 5] - u[:, 10], "whole tile": u[:, 5] - u[:, 0]}
for k, v in ph.items():
    print(f"   {k:28s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f} us")

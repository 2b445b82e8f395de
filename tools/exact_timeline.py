"""Per-workgroup phase timeline of k_exact (diagnostic build: make -C optimalmodulationds_amd/csrc timeline).

The last launch that runs pass1_tile before the fetch is the k_exact of the last horizon step (audit and sweep switched
off).  Stamps (s_memrealtime, 100 MHz): 0 entry, 8 layer-1 loads back, 9 LDS written, 1 layer-1 tile built, 6 first GEMM
done, 2..4 hidden layers done, 10 last-layer MFMA loop done, 5 exit, plus HW_ID / XCC_ID.
usage: python tools/exact_timeline.py [rollouts]
"""
import ctypes as C
import os, sys
import numpy as np
os.environ.setdefault("OMDS_SCREEN_AUDIT", "0")
os.environ.setdefault("OMDS_SCREEN_SWEEP", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optimalmodulationds_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "optimalmodulationds_amd", "csrc", "libomds_hip_tl.so")
from optimalmodulationds_amd import scenes
from optimalmodulationds_amd.engine import Engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H, k, K = 32, 5, 6
z = np.load(os.path.join(ROOT, "tests", "golden", "weights", "franka.npz"))
W = [z[f"W{i}"] for i in range(5)]; b = [z[f"b{i}"] for i in range(5)]
obs, q0, qf = scenes.shelf_scene(), scenes.FRANKA_Q0, scenes.FRANKA_QF
eng = Engine(7, N, H, k, max_obs=max(64, obs.shape[0]))
eng.set_mlp(W, b); eng.set_obstacles(obs)
eng.params.dt, eng.params.dst_thr, eng.params.ignored_links = 0.5, 0.01, 0b111
eng.push_params(); eng.set_ds(qf)
rng = np.random.RandomState(1234)
s = (np.arange(K) + 0.5) / K
mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
sg_c, al_c = np.ones(K, np.float32), rng.standard_normal((K, 7)).astype(np.float32)
for it in range(4):
    eng.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=1234 + it)
    eng.propagate(q0)
st = eng.screen_stats()
nwg = 768
buf = np.zeros((nwg, 16), dtype=np.uint64)
fn = eng.lib.omds_timeline_fetch
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert fn(eng.h, buf.ctypes.data, nwg) == 0
eng.close()
t = buf[:, :12].astype(np.int64)
live = (t[:, 0] > 0) & (t[:, 5] > t[:, 0]) & (t[:, 1] > 0)
live &= t[:, 0] > t[live, 0].max() - 10000   # the last launch only: workgroups without a tile keep the stamps of an earlier one
hw = buf[:, 7]
t0 = t[live, 0].min()
us = (t - t0) / 100.0
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
h = hw.astype(np.int64) & 0xFFFFFFFF
slot = ((xcc * 8 + ((h >> 13) & 7)) * 2 + ((h >> 12) & 1)) * 16 + ((h >> 8) & 0xF)
print(f"screening {st['candidates_per_rollout_step']:.2f} candidates per rollout-step; workgroups with a tile {live.sum()} of {nwg}, "
      f"on {len(np.unique(slot[live]))} CUs; last exit {us[live, 5].max():.1f} us after the first entry")
cnt = np.bincount(np.unique(slot[live], return_counts=True)[1])
print("tiles per CU:", {i: int(c) for i, c in enumerate(cnt) if c})
u = us[live]
ph = {"entry (after first)": u[:, 0], "L1: loads back": u[:, 8] - u[:, 0], "L1: LDS written": u[:, 9] - u[:, 8], "L1: barrier": u[:, 1] - u[:, 9],
      "GEMM 1": u[:, 6] - u[:, 1], "epilogue 1": u[:, 2] - u[:, 6], "layer 2": u[:, 3] - u[:, 2], "layer 3": u[:, 4] - u[:, 3],
      "last: MFMA loop": u[:, 10] - u[:, 4], "last: rest": u[:, 5] - u[:, 10], "whole tile": u[:, 5] - u[:, 0], "exit (after first entry)": u[:, 5]}
print("phase: mean / p10 / p90 / max us")
for kk, v in ph.items():
    print(f"   {kk:26s} {v.mean():7.2f} {np.percentile(v, 10):7.2f} {np.percentile(v, 90):7.2f} {v.max():7.2f}")

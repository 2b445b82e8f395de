"""Per-workgroup phase timeline of k_pass1 (diagnostic build: make -C optimalmodulationds_amd/csrc timeline).

Each workgroup of the last k_pass1 launch records s_memrealtime (100 MHz) at: 0 entry, 1 layer-1 tile built,
6 first GEMM done, 2..4 hidden layers done, 5 exit, plus HW_ID / XCC_ID.  Prints where a tile's time goes, how long a
CU slot stays empty between two workgroups, and the ramp / drain of the launch.
usage: python tools/pass1_timeline.py [rollouts]
"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optimalmodulationds_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "optimalmodulationds_amd", "csrc", "libomds_hip_tl.so")
from optimalmodulationds_amd import scenes
from optimalmodulationds_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
z = np.load(os.path.join(ROOT, "tests", "golden", "weights", "franka.npz"))
W = [z[f"W{i}"] for i in range(5)]; b = [z[f"b{i}"] for i in range(5)]
eng = Engine(7, B, 1, 5, max_obs=512)
eng.set_mlp(W, b); obs = scenes.shelf_scene(); eng.set_obstacles(obs)
q = (scenes.FRANKA_Q0 + 0.3 * np.random.RandomState(0).standard_normal((B, 7))).astype(np.float32)
for _ in range(4):
    eng.dist_grad(q)
total = B * obs.shape[0]
tiles64 = total // 64
n_big = tiles64 - 256 if total >= 64 * 1024 else 0
nwg = n_big + (total - n_big * 64 + 31) // 32 if n_big else (total + 31) // 32
buf = np.zeros((nwg, 16), dtype=np.uint64)
fn = eng.lib.omds_timeline_fetch
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert fn(eng.h, buf.ctypes.data, nwg) == 0
eng.close()
t = buf[:, :12].astype(np.int64)
hw = buf[:, 7]
t0 = t[:, 0].min()
us = (t - t0) / 100.0            # 100 MHz -> microseconds
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
h = hw.astype(np.int64) & 0xFFFFFFFF
cu = (h >> 8) & 0xF; sh = (h >> 12) & 0x1; se = (h >> 13) & 0x7
slot = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(f"rows {total}, workgroups {nwg} ({n_big} x 64 rows), launch span {us[:, 5].max():.1f} us, distinct CUs {len(np.unique(slot))}")
big = np.arange(nwg) < n_big
for name, sel in (("64-row", big), ("32-row", ~big)):
    if sel.sum() == 0:
        continue
    u = us[sel]
    ph = {"L1: loads back": u[:, 8] - u[:, 0], "L1: LDS written": u[:, 9] - u[:, 8], "L1: barrier": u[:, 1] - u[:, 9],
          "last: MFMA loop": u[:, 10] - u[:, 4], "last: rest": u[:, 5] - u[:, 10],
          "layer-1 build": u[:, 1] - u[:, 0], "GEMM 1": u[:, 6] - u[:, 1], "epilogue 1": u[:, 2] - u[:, 6],
          "layer 2": u[:, 3] - u[:, 2], "layer 3": u[:, 4] - u[:, 3], "last layer": u[:, 5] - u[:, 4], "whole tile": u[:, 5] - u[:, 0]}
    print(f"-- {name} tiles ({sel.sum()}): mean / p10 / p90 us")
    for k, v in ph.items():
        print(f"   {k:14s} {v.mean():7.2f} {np.percentile(v, 10):7.2f} {np.percentile(v, 90):7.2f}")
# per CU: occupancy over time and gaps between successive workgroups
gaps = []
conc = np.zeros(int(us[:, 5].max()) + 2)
for s in np.unique(slot):
    idx = np.where(slot == s)[0]
    ev = sorted([(us[i, 0], +1) for i in idx] + [(us[i, 5], -1) for i in idx])
    # time with fewer than 2 resident workgroups between first start and last end on this CU
    cur = 0; last = ev[0][0]; under = 0.0
    for (tt, d) in ev:
        if cur < 2:
            under += tt - last
        cur += d; last = tt
    gaps.append((under, ev[-1][0] - ev[0][0], ev[0][0], ev[-1][0]))
g = np.array(gaps)
print(f"per CU: first start {g[:, 2].mean():.1f} us (max {g[:, 2].max():.1f}), last end mean {g[:, 3].mean():.1f} min {g[:, 3].min():.1f} max {g[:, 3].max():.1f}")
print(f"per CU: time with < 2 resident workgroups {g[:, 0].mean():.1f} us of {g[:, 1].mean():.1f} us busy span")
# dispatch latency: for each workgroup end, the next start on the same CU
lat = []
for s in np.unique(slot):
    idx = np.where(slot == s)[0]
    ends = np.sort(us[idx, 5]); starts = np.sort(us[idx, 0])
    for e in ends:
        nx = starts[starts >= e - 0.005]
        if len(nx):
            lat.append(nx[0] - e)
lat = np.array(lat)
print(f"slot refill latency (end -> next start on that CU): mean {lat.mean():.2f} us, p50 {np.percentile(lat, 50):.2f}, p90 {np.percentile(lat, 90):.2f}")
# matrix-pipe coverage per CU: time during which 0 / 1 / 2 resident workgroups are inside a GEMM loop
EPI = 0.7   # epilogue + barriers per hidden layer, from the layer-1 split above
cov = np.zeros(3); span_tot = 0.0
for s_ in np.unique(slot):
    idx = np.where(slot == s_)[0]
    ev = []
    for i in idx:
        for a, b_ in ((us[i, 1], us[i, 6]), (us[i, 2], us[i, 3] - EPI), (us[i, 3], us[i, 4] - EPI)):
            ev.append((a, +1)); ev.append((b_, -1))
    ev.sort()
    cur = 0; last = us[idx, 0].min()
    for (tt, d) in ev:
        cov[min(cur, 2)] += tt - last
        cur += d; last = tt
    cov[0] += us[idx, 5].max() - last
    span_tot += us[idx, 5].max() - us[idx, 0].min()
print("share of each CU's busy span with 0 / 1 / 2 workgroups inside a GEMM loop: " + " / ".join(f"{100 * c / span_tot:.1f} %" for c in cov))
print(f"mean CU busy span {span_tot / len(np.unique(slot)):.1f} us of launch span {us[:, 5].max():.1f} us")
wpc = np.bincount(slot)[np.unique(slot)]
print(f"workgroups per CU: min {wpc.min()} max {wpc.max()}")

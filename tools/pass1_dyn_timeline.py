"""Per-workgroup phase timeline of the compacted k_pass1 (k_pass1_dyn; diagnostic build: make -C optimalmodulationds_amd/csrc timeline).

Thread 0 of every workgroup of the last launch records wall_clock64 (100 MHz) at: 0 entry, 1 inputs gathered, 2 + 2k its wave's share of
product k done (k = 0: layer 1, 1..3: the hidden layers), 3 + 2k level k stored (behind the barrier), 10 exit, 15 HW_ID / XCC_ID.  Prints
where a tile's time goes and, per CU, the share of the busy span with 0 / 1 / 2 resident workgroups inside a product.
usage: python tools/pass1_dyn_timeline.py [rollouts]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optimalmodulationds_amd._lib as L   # noqa: E402
L.LIB_PATH = os.path.join(ROOT, "optimalmodulationds_amd", "csrc", "libomds_hip_tl.so")
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
buf = np.zeros((nwg, 16), dtype=np.uint64)
fn = eng.lib.omds_timeline_fetch
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert fn(eng.h, buf.ctypes.data, nwg) == 0
print(eng.pass1_skip_stats())
eng.close()
t = buf[:, :15].astype(np.int64)
hw = buf[:, 15]
us = (t - t[:, 0].min()) / 100.0
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
h = hw.astype(np.int64) & 0xFFFFFFFF
cu = (h >> 8) & 0xF; sh = (h >> 12) & 0x1; se = (h >> 13) & 0x7
slot = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(f"rows {total}, workgroups {nwg} ({n_big} x 64 rows), launch span {us[:, 10].max():.1f} us, distinct CUs {len(np.unique(slot))}")
big = np.arange(nwg) < n_big
for name, sel in (("64-row", big), ("32-row", ~big)):
    u = us[sel]
    print(f"-- {name} tiles ({sel.sum()}): mean / p10 / p90 us")
    ph = [("gather", u[:, 1] - u[:, 0])]
    for k in range(4):
        ph.append((f"product {k}", u[:, 2 + 2 * k] - (u[:, 1] if k == 0 else u[:, 1 + 2 * k])))
        ph.append((f"store level {k}", u[:, 3 + 2 * k] - u[:, 2 + 2 * k]))
    for k in range(1, 4):
        ok = t[sel][:, 10 + k] > 0
        if ok.any():
            ph.append((f"  product {k}: operands of chunk 0 here", (us[sel][ok][:, 10 + k] - u[ok][:, 1 + 2 * k])))
    ph.append(("last layer", u[:, 10] - u[:, 9]))
    ph.append(("whole tile", u[:, 10] - u[:, 0]))
    for kname, v in ph:
        print(f"   {kname:16s} {v.mean():7.2f} {np.percentile(v, 10):7.2f} {np.percentile(v, 90):7.2f}")
cov = np.zeros(3); span_tot = 0.0
for s_ in np.unique(slot):
    idx = np.where(slot == s_)[0]
    ev = []
    for i in idx:
        for k in range(4):
            a = us[i, 1] if k == 0 else us[i, 1 + 2 * k]
            ev.append((a, +1)); ev.append((us[i, 2 + 2 * k], -1))
    ev.sort()
    cur = 0; last = us[idx, 0].min()
    for (tt, d) in ev:
        cov[min(cur, 2)] += tt - last
        cur += d; last = tt
    cov[0] += us[idx, 10].max() - last
    span_tot += us[idx, 10].max() - us[idx, 0].min()
print("share of each CU's busy span with 0 / 1 / 2 workgroups inside a product: " + " / ".join(f"{100 * c / span_tot:.1f} %" for c in cov))
res = np.zeros(3); launch_span = us[:, 10].max() - us[:, 0].min(); ncu = len(np.unique(slot))
for s_ in np.unique(slot):   # resident workgroups (between a workgroup's first and last stamp) per CU over the LAUNCH span
    idx = np.where(slot == s_)[0]
    ev = sorted([(us[i, 0], +1) for i in idx] + [(us[i, 10], -1) for i in idx])
    cur = 0; last = us[:, 0].min()
    for (tt, d) in ev:
        res[min(cur, 2)] += tt - last
        cur += d; last = tt
    res[0] += us[:, 10].max() - last
print("share of the launch span with 0 / 1 / 2 workgroups resident on a CU: " + " / ".join(f"{100 * c / (launch_span * ncu):.1f} %" for c in res))
print(f"mean CU busy span {span_tot / len(np.unique(slot)):.1f} us of launch span {us[:, 10].max():.1f} us")

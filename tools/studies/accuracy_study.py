#!/usr/bin/env python3
"""How far apart may two CORRECT fp32 evaluations of the distance network be?  (CPU only, reads the committed fixtures.)

For the closest-obstacle rows of the reference-captured states (`st_q` of the fixtures) the network is evaluated
  truth   : float64 throughout
  f64acc  : fp32 activations, every dot product accumulated in float64 and rounded once (the best any fp32 pipeline can do)
  chain   : fp32 fmaf chain in k order starting at the bias = what v_mfma_f32_32x32x2_f32 computes (k_pass1 / pass 2)
  blas    : numpy's BLAS sgemm (the numpy oracle)
and compared with the float64 result and with the distance the REFERENCE (torch CPU) recorded in the fixture.  The
reference itself is ~3e-7 x scale away from exact arithmetic -- as far as the MFMA chain is -- so no choice of summation
order on the GPU can bring the two closer than that; MPPI.py:149-155 then multiplies the difference by a sigmoid slope of
up to 100.  tests/test_gpu_parity.py (stage C) therefore compares modulated velocities inside the envelope spanned by
+-DIST_ULP around the reference's distance instead of loosening the tolerance on the velocity itself."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from helpers import load, weights_path  # noqa: E402
from oracle import omds_oracle as orc  # noqa: E402

F32, F64 = np.float32, np.float64


def forward(m, x, mode):
    h = orc.positional_encoding(x)
    for i in range(len(m.W)):
        W, b = m.W[i], m.b[i]
        if mode == "truth":
            z = h.astype(F64) @ W.T.astype(F64) + b.astype(F64)
        elif mode == "f64acc":
            z = (h.astype(F64) @ W.T.astype(F64) + b.astype(F64)).astype(F32)
        elif mode == "chain":
            acc = np.broadcast_to(b, (h.shape[0], W.shape[0])).astype(F32).copy()
            for k in range(W.shape[1]):      # one rounding per fused multiply-add
                acc = (acc.astype(F64) + h[:, k:k + 1].astype(F64) * W[:, k][None, :].astype(F64)).astype(F32)
            z = acc
        else:
            z = (h @ W.T + b).astype(F32)
        h = np.maximum(z, 0) if i < len(m.W) - 1 else z
        if mode != "truth":
            h = h.astype(F32)
    return h


print(f"{'fixture':26s} {'max |d|':>8s}   max |x - float64| for x = f64acc / chain / blas / REFERENCE      max |x - reference| for x = f64acc / chain / blas")
for name, kind in [("franka_shelf_K6", "franka"), ("franka_shelf_collide_K4", "franka"), ("franka_cross_K3", "franka"),
                   ("planar7_K4", "planar7"), ("planar2_c1_K3", "planar2")]:
    fx = load(name)
    m = orc.Mlp.from_npz(weights_path(kind))
    q, obs, k = fx["st_q"], fx["obs"], int(fx["k"])
    _, _, _, idx = orc.distance_repulsion_nn(m, q, obs, k, fx["ignored_links"])
    rows = np.hstack([q, obs[idx[:, 0], :3]]).astype(F32)[:, :m.W[0].shape[1] // 3]
    div = 100.0 if m.out_channels == 9 else 1.0
    res = {}
    for mode in ("truth", "f64acc", "chain", "blas"):
        y = forward(m, rows, mode)
        res[mode] = y[np.arange(len(y)), np.argmin(y, 1)] / div - obs[idx[:, 0], 3]
    ref = fx["st_distance"].astype(F64)
    e = lambda a, b: float(np.abs(a - b).max())  # noqa: E731
    print(f"{name:26s} {np.abs(ref).max():8.3f}   {e(res['f64acc'], res['truth']):.2e} / {e(res['chain'], res['truth']):.2e} / "
          f"{e(res['blas'], res['truth']):.2e} / {e(ref, res['truth']):.2e}          "
          f"{e(res['f64acc'], ref):.2e} / {e(res['chain'], ref):.2e} / {e(res['blas'], ref):.2e}")

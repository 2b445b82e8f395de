"""CPU study behind include/omds.h's screening contract: is there a usable A-PRIORI bound on |Da - D|, the error of the f16
screening pass against the fp32 network?  Propagates the f16 rounding of inputs, weights and activations (2^-11 relative each; the
fp32 accumulation is negligible beside it) through sum |W| per layer -- (a) rigorously, with interval bounds on the activations from
|q| <= Qmax, |p| <= Pmax; (b) to first order, with the activation magnitudes of 20 000 sampled inputs -- and prints both beside the
measured bound eps the library calibrates (bench.py: screening.eps).  Result: 3e4 times too large to select anything; the library's
default is therefore the all-fp32 step and screening is opt-in.   python tools/studies/screen_apriori_bound.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import omds_oracle as orc   # noqa: E402

U = 2.0 ** -11
for kind, Qmax, Pmax, div, eps in (("franka", 2.9, 1.5, 100.0, 0.016), ("planar7", 3.2, 8.0, 1.0, None)):
    m = orc.Mlp.from_npz(os.path.join(ROOT, "tests", "golden", "weights", kind + ".npz"))
    n = m.W[0].shape[1] // 3 - 3
    F = np.concatenate([np.r_[np.full(n, Qmax), np.full(3, Pmax)], np.ones(n + 3), np.ones(n + 3)])
    H, e = F.copy(), U * F
    for l, (W, b) in enumerate(zip(m.W, m.b)):
        A = np.abs(W).astype(np.float64)
        ez = A @ e + U * (A @ (H + e))
        Hn = A @ H + np.abs(b)
        if l < len(m.W) - 1:
            e = ez * (1 + U) + U * Hn
        H = Hn
    print(f"{kind}: rigorous a-priori bound on |Da - D| = {ez.max() / div:.4g} (interval bound on the network output itself: {H.max() / div:.4g})")
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-Qmax, Qmax, (20000, n)), rng.uniform(-Pmax, Pmax, (20000, 3))], 1).astype(np.float32)
    h = orc.positional_encoding(x).astype(np.float64)
    e = U * np.abs(h)
    for l, (W, b) in enumerate(zip(m.W, m.b)):
        A = np.abs(W).astype(np.float64)
        ez = e @ A.T + U * ((np.abs(h) + e) @ A.T)
        z = h @ W.T.astype(np.float64) + b
        if l < len(m.W) - 1:
            h = np.maximum(z, 0)
            e = ez * (1 + U) + U * np.abs(h)
    print(f"{kind}: first-order bound with sampled activation magnitudes: max {ez.max() / div:.4g}, median over rows {np.median(ez.max(axis=1)) / div:.4g}"
          + (f"; measured eps of the shelf scene {eps} -> ratio {np.median(ez.max(axis=1)) / div / eps:.3g}" if eps else ""))

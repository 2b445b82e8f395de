"""GPU study: how close to BIT-identical are the device's network outputs to the oracle's (= the reference's arithmetic restated)?
Per scenario fixture: the pass-1 min-distance matrix, the selected distance, the blended gradient, and raw MLP outputs / vjp.
    python tools/studies/device_bits_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import MLP_KINDS, SCENARIOS, load, weights_path   # noqa: E402
from oracle import omds_oracle as orc                          # noqa: E402
from test_gpu_parity import _engine                            # noqa: E402
from optimalmodulationds_amd.engine import Engine              # noqa: E402


def ulps(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia); ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


for kind in MLP_KINDS:
    fx = load("mlp_" + kind)
    m = orc.Mlp.from_npz(weights_path(kind))
    eng = Engine(fx["x"].shape[1] - 3, 128, 1, 1, max_obs=8)
    eng.set_mlp(m.W, m.b, act=m.act, skip_after=m.skip_after)
    y, g, mi = eng.mlp_forward_vjp(fx["x"])
    yo, go, mio = orc.mlp_vjp_argmin(m, fx["x"])
    print(f"mlp_{kind:12s} forward identical bits {np.mean(y == yo):.4f} (max {ulps(y, yo).max()} ulp)  vs reference {np.mean(y == fx['y']):.4f}   "
          f"grad identical {np.mean(g == go):.4f}, max |dg| / max|g| {np.abs(g - go).max() / np.abs(go).max():.1e}")
    eng.close()
for name in SCENARIOS:
    fx = load(name)
    eng, m = _engine(fx)
    d, g, mind, idx = eng.dist_grad(fx["st_q"], want_mindist=True, want_idx=True)
    do, go, mo, io = orc.distance_repulsion_nn(m, fx["st_q"], fx["obs"], int(fx["k"]), fx["ignored_links"])
    print(f"{name:28s} mindist identical {np.mean(mind == mo):.5f} (max {ulps(mind, mo).max()} ulp; vs reference {np.mean(mind == fx['st_mindist']):.5f})  "
          f"idx same {np.mean(idx == io):.4f}  distance identical {np.mean(d == do):.4f} (vs ref {np.mean(d == fx['st_distance']):.4f})  "
          f"grad max rel {np.abs(g - go).max() / np.abs(go).max():.1e} (vs ref {np.abs(g - fx['st_nn_grad']).max() / np.abs(go).max():.1e})")
    eng.close()

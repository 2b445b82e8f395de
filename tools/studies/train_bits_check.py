"""The trainer's special-shape kernels (k_gemm_tall, k_gemm_thin, k_gemm_thin_out, k_wgrad_thin) against the general k_gemm: STEPS training steps
from the same start, weights and Adam state dumped; run twice (experiment build: OMDS_TALL_DBG=24 = the general kernel everywhere, then 0) and
compare the dumps for array equality.   python tools/studies/train_bits_check.py out.npz [relu|tanh] [rows]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import omds_oracle as orc                      # noqa: E402  (the weights loader only)
from optimalmodulationds_amd.trainer import SdfTrainer     # noqa: E402

out, act, B = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "relu"), int(sys.argv[3]) if len(sys.argv) > 3 else 20011
m = orc.Mlp.from_npz(os.path.join(ROOT, "tests", "golden", "weights", "franka.npz"))
rng = np.random.RandomState(4)
x = rng.uniform(-2.0, 2.0, (B, 10)).astype(np.float32)
y = (orc.mlp_forward(m, x) + 5.0 * rng.standard_normal((B, 9))).astype(np.float32)
tr = SdfTrainer([30, 256, 256, 256, 256, 9], act)
tr.set_weights(m.W, m.b)
tr.set_data(x, y)
losses = [tr.step(lr=2e-4) for _ in range(6)]
W, b = tr.get_weights()
st = tr.optimizer_state_dict()["state"]
np.savez(out, losses=np.array(losses, np.float32), **{f"W{i}": w for i, w in enumerate(W)}, **{f"b{i}": v for i, v in enumerate(b)},
         **{f"m{i}": st[i]["exp_avg"].numpy() for i in st}, **{f"v{i}": st[i]["exp_avg_sq"].numpy() for i in st})
print("losses", losses)

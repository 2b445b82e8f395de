"""Gradients of one trainer step against torch-CPU autograd, per parameter tensor (30-256^4-9 from the shipped Franka weights, ReLU,
3001 rows): Adam's exp_avg after the first step is (1 - beta1) * g, so the optimizer state hands back the gradient the device
computed.  Prints, per tensor: max |g_dev - g_torch| / max |g_torch|, the share of elements whose torch gradient is exactly zero,
the share of those that are non-zero on the device, and sign disagreements among the rest.  (Adam moves a weight by lr * sign(g) at
its first steps whatever |g| is: an element that is 0 in one run and +-1e-12 in the other ends 2e-4 apart.)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import omds_oracle as orc                      # noqa: E402  (the weights loader only)
from optimalmodulationds_amd.trainer import SdfTrainer     # noqa: E402

act = sys.argv[1] if len(sys.argv) > 1 else "relu"
m = orc.Mlp.from_npz(os.path.join(ROOT, "tests", "golden", "weights", "franka.npz"))
rng = np.random.RandomState(4)
B = 3001
x = rng.uniform(-2.0, 2.0, (B, 10)).astype(np.float32)
y = (orc.mlp_forward(m, x) + 5.0 * rng.standard_normal((B, 9))).astype(np.float32)
tr = SdfTrainer([30, 256, 256, 256, 256, 9], act)
tr.set_weights(m.W, m.b)
tr.set_data(x, y)
Wt = [torch.tensor(w.copy(), requires_grad=True) for w in m.W]
bt = [torch.tensor(v.copy(), requires_grad=True) for v in m.b]
xt, yt = torch.from_numpy(x), torch.from_numpy(y)
h = torch.cat((xt, torch.sin(xt), torch.cos(xt)), dim=1)
for i in range(5):
    h = F.linear(h, Wt[i], bt[i])
    if i < 4:
        h = torch.relu(h) if act == "relu" else torch.tanh(h)
loss = F.mse_loss(h, yt)
loss.backward()
got = tr.step(lr=2e-4)
st = tr.optimizer_state_dict()["state"]
print(f"loss device {got:.7f} torch {loss.item():.7f}")
for i in range(5):
    for j, (name, p) in enumerate((("W", Wt[i]), ("b", bt[i]))):
        g_t = p.grad.numpy()
        g_d = st[2 * i + j]["exp_avg"].numpy() / 0.1
        sc = float(np.abs(g_t).max())
        z = g_t == 0
        nz_dev = (g_d[z] != 0).mean() if z.any() else 0.0
        rest = ~z
        flips = (np.sign(g_d[rest]) != np.sign(g_t[rest])).mean() if rest.any() else 0.0
        print(f"{name}{i}: max|diff|/max|g| {np.abs(g_d - g_t).max() / sc:.2e}  torch-zero {z.mean():.4f}  of them non-zero on the device {nz_dev:.4f}"
              f"  (largest {np.abs(g_d[z]).max() / sc if z.any() else 0:.1e} x max|g|)  sign flips among the rest {flips:.5f}")

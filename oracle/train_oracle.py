"""CPU restatement (numpy, fp32) of one epoch of the reference's SDF training -- TEST INFRASTRUCTURE ONLY, like omds_oracle.py:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product path (csrc/train.hip) never does.

What it restates (paths relative to /root/reference/python_scripts/mlp_learn/):
  * train_sdf.py:105-113   full-batch forward, F.mse_loss(reduction='mean'), backward, optimizer.step() with
                           torch.optim.Adam(lr=2e-4) (train_sdf.py:84); autocast / GradScaler are no-ops on the CPU;
  * sdf/network_macros_mod.py:137-146   MLPRegression.forward with NeRF features [x, sin x, cos x], skips = [];
  * torch.optim.Adam's single-tensor update (torch 2.10, the pinned third-party arithmetic): exp_avg.lerp_(grad, 1 - beta1);
    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2); denom = exp_avg_sq.sqrt() / sqrt(1 - beta2^t) + eps;
    param.addcdiv_(exp_avg, denom, value = -lr / (1 - beta1^t)).
Pinned by tests/test_oracle_golden.py::test_train_oracle_matches_the_reference_run against tests/golden/train_sdf_planar2.npz
(tools/make_golden_train.py: the reference's own model class and torch's Adam, 100 epochs)."""
import numpy as np

F32 = np.float32


class TrainState:
    def __init__(self, W, b, act="relu"):
        self.W = [np.array(w, dtype=F32) for w in W]
        self.b = [np.array(v, dtype=F32) for v in b]
        self.act = act
        self.m = [np.zeros_like(w) for w in self.W] + [np.zeros_like(v) for v in self.b]
        self.v = [np.zeros_like(w) for w in self.W] + [np.zeros_like(v) for v in self.b]
        self.t = 0


def forward(st, x):
    x = np.asarray(x, dtype=F32)
    H = [np.concatenate((x, np.sin(x), np.cos(x)), axis=1).astype(F32)]          # network_macros_mod.py:139-140
    L = len(st.W)
    for i in range(L):
        z = (H[-1] @ st.W[i].T + st.b[i]).astype(F32)
        H.append((np.maximum(z, F32(0)) if st.act == "relu" else np.tanh(z)).astype(F32) if i + 1 < L else z)
    return H


def mse(pred, y):
    return float(np.mean((pred.astype(np.float64) - y.astype(np.float64)) ** 2))


def train_step(st, x, y, lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8):
    """One epoch (train_sdf.py:105-113).  Returns the loss before the update."""
    y = np.asarray(y, dtype=F32)
    H = forward(st, x)
    L = len(st.W)
    loss = mse(H[L], y)
    G = (F32(2.0 / y.size) * (H[L] - y)).astype(F32)
    gW, gb = [None] * L, [None] * L
    for i in range(L - 1, -1, -1):
        gW[i] = (G.T @ H[i]).astype(F32)
        gb[i] = G.sum(axis=0).astype(F32)
        if i > 0:
            G = (G @ st.W[i]).astype(F32)
            h = H[i]
            G = (G * ((h > 0).astype(F32) if st.act == "relu" else (F32(1) - h * h))).astype(F32)
    st.t += 1
    bc1, bc2 = 1.0 - beta1 ** st.t, 1.0 - beta2 ** st.t
    step_size, bc2_sqrt = F32(lr / bc1), F32(np.sqrt(bc2))
    params, grads = st.W + st.b, gW + gb
    for j, (p, g) in enumerate(zip(params, grads)):
        st.m[j] = (st.m[j] + (g - st.m[j]) * F32(1 - beta1)).astype(F32)
        st.v[j] = (st.v[j] * F32(beta2) + F32(1 - beta2) * g * g).astype(F32)
        denom = (np.sqrt(st.v[j]) / bc2_sqrt + F32(eps)).astype(F32)
        p -= (step_size * (st.m[j] / denom)).astype(F32)
    return loss

"""Would a spatial order of the OBSTACLES let k_screen skip more? (EXPERIMENTS.md)

A wave of k_screen multiplies 32 consecutive pairs t * O + o: one rollout (or the end of one and the start of the next) x 32
consecutive obstacles.  A k-chunk of 16 hidden units is skipped when all 16 are zero for all 32 pairs (exact).  The
hidden units are already sorted by firing frequency (screen_reorder); this study asks what the order of the obstacle axis adds:
natural (the scene's own construction order: runs of 12 spheres along one shelf board), Morton (z-order of the sphere centres),
k-means clusters of 32, a 1-D sort along the principal axis, and a random permutation as the floor.  States: real rollouts of the
shelf task from the oracle (N rollouts x H steps, noise-driven policy), so that consecutive rollouts at one horizon step are as
similar as they are in a run.  Per order: the fraction of the kernel's 416 MFMAs per wave that would not be issued -- with the
shipped kernel's rule (only chunks >= 10 are tested) and with every chunk tested (the ceiling).  numpy on the oracle:

    python tools/studies/pair_coherence.py [N] [H]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import omds_oracle as orc          # noqa: E402
from optimalmodulationds_amd import scenes     # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
m = orc.Mlp.from_npz(os.path.join(ROOT, "tests", "golden", "weights", "franka.npz"))
obs = scenes.shelf_scene()
O = obs.shape[0]
rng = np.random.RandomState(0)
q0, qf = np.array(scenes.FRANKA_Q0, np.float32), np.array(scenes.FRANKA_QF, np.float32)
K = 10
s = (np.arange(K) + 0.5) / K
mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
mu = np.repeat(mu_c[None], N, 0)
sg = np.ones((N, K), np.float32)
al = (rng.standard_normal((K, 7)) + 3.0 * rng.standard_normal((N, K, 7))).astype(np.float32)
out = orc.propagate(m, q0, qf, obs, N=N, H=H, dt=0.5, k=5, ignored_links=[0, 1, 2], mu_tmp=mu, sigma_tmp=sg, alpha_tmp=al, prm=orc.Params(dst_thr=0.01))
traj = out.all_traj                                # [N, H, 7]


def masks_for(states):
    """[T, O, 4, 256] bool: hidden unit alive per (state, obstacle, layer)."""
    T = states.shape[0]
    A = np.zeros((T, O, 4, 256), bool)
    for t in range(T):
        x = np.concatenate([np.repeat(states[t:t + 1], O, 0), obs[:, :3]], 1).astype(np.float32)
        h = orc.positional_encoding(x)
        for i in range(4):
            h = np.maximum(h @ m.W[i].T + m.b[i], 0)
            A[t, :, i] = h > 0
    return A


def morton(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    g = ((p - lo) / np.maximum(hi - lo, 1e-9) * ((1 << bits) - 1)).astype(np.int64)
    code = np.zeros(len(p), np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((g[:, a] >> b) & 1) << (3 * b + a)
    return np.argsort(code, kind="stable")


def kmeans_order(p, size=32, iters=30):
    k = int(np.ceil(len(p) / size))
    c = p[rng.choice(len(p), k, replace=False)]
    for _ in range(iters):
        a = np.argmin(((p[:, None] - c[None]) ** 2).sum(-1), 1)
        c = np.stack([p[a == j].mean(0) if (a == j).any() else c[j] for j in range(k)])
    return np.lexsort((np.arange(len(p)), a))


def pca_order(p):
    x = p - p.mean(0)
    w = np.linalg.svd(x, full_matrices=False)[2][0]
    return np.argsort(x @ w, kind="stable")


ORDERS = {"natural": np.arange(O), "morton": morton(obs[:, :3]), "kmeans32": kmeans_order(obs[:, :3]), "pca-1d": pca_order(obs[:, :3]),
          "random": rng.permutation(O)}
W_SAVED = np.array([8, 8, 8, 1])       # MFMAs a dead chunk of layer L's output saves in the layer behind it (8 row slices; the last layer has 1)
TOTAL = 416                            # MFMAs per wave and tile: 16 (layer 1) + 3 * 128 + 16

steps = [1, H // 2, H - 1]
print(f"shelf scene O = {O}, {N} rollouts x {H} steps from the oracle; horizon steps sampled: {steps}")
print(f"{'obstacle order':12s} {'step':>4s}  " + "  ".join(f"dead L{i + 1}" for i in range(4)) + "   skipped (chunks >= 10 tested)   skipped (all chunks tested)")
summary = {}
for h in steps:
    A = masks_for(traj[:, h])          # rollouts at one horizon step, in rollout order: what one launch sees
    for name, perm in ORDERS.items():
        P = A[:, perm].reshape(N * O, 4, 256)
        nb = (N * O) // 32
        blk = P[:nb * 32].reshape(nb, 32, 4, 256).any(1)           # [nb, 4, 256] unit alive for the wave
        dead_t, dead_a, row = 0.0, 0.0, []
        for L in range(4):
            order = np.argsort(-blk[:, L].mean(0), kind="stable")    # the library's sort: most frequently alive first
            ch = blk[:, L][:, order].reshape(nb, 16, 16).any(2)      # [nb, 16 chunks]
            row.append(1 - ch.mean())
            dead_t += W_SAVED[L] * (~ch[:, 10:]).sum(1).mean()
            dead_a += W_SAVED[L] * (~ch).sum(1).mean()
        summary.setdefault(name, []).append((dead_t / TOTAL, dead_a / TOTAL))
        print(f"{name:12s} {h:4d}  " + "  ".join(f"{x:7.3f}" for x in row) + f"   {100 * dead_t / TOTAL:10.1f} %                    {100 * dead_a / TOTAL:10.1f} %")
print()
for name, v in summary.items():
    v = np.array(v)
    print(f"{name:12s} mean over the steps: {100 * v[:, 0].mean():5.1f} % of the MFMAs skipped with the shipped test rule, {100 * v[:, 1].mean():5.1f} % with every chunk tested")
print("build criterion (VERDICT r04 3a): >= 35 % of the hidden-layer MFMAs")

"""CPU study: which association of the pass-2 forward meets north_star's PLAIN 1e-5 bar against the reference's own steps.

Teacher-forced over the committed reference fixtures (tests/golden/*.npz, made by tools/make_golden.py from /root/reference): every
horizon step restarts from the reference's state, the oracle does everything (pass 1, selection, backward, modulation) EXCEPT the
forward of the N k selected rows, which is recomputed in the arithmetic under study; the resulting modulated velocity is compared
with the reference's (q_next - q) / dt by tests/helpers.plain_bar (every row, |u - u_ref| <= 1e-5 max|u_ref|, no envelope).

fp32 fmaf is emulated through float64 (a product of two floats is exact in a double; the sum is rounded to 53, then to 24 bits: a
double rounding in ~2^-29 of the cases).

Finding that fixes the device's arithmetic (section "torch" below): for M >= 11 rows torch-CPU's addmm (MKL sgemm, AVX-512) IS, bit for
bit, one fmaf chain per output element in ASCENDING k from zero with the bias added afterwards.

    python tools/studies/assoc_order_study.py [--torch]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load, plain_bar, seds_of, weights_path, SCENARIOS   # noqa: E402
from oracle import omds_oracle as orc                                   # noqa: E402

F32 = np.float32
DEV8 = (0, 4, 1, 5, 2, 6, 3, 7)          # k order of the round-5 device inside a chunk of 8 (gemm256)


def chain(x, W, b, order=None, bias_first=False):
    """[B, K] . [out, K]^T as one fmaf chain per output element in the given k order."""
    B, K = x.shape
    acc = np.broadcast_to(b, (B, W.shape[0])).astype(F32).copy() if bias_first else np.zeros((B, W.shape[0]), F32)
    for k in (range(K) if order is None else order):
        acc = (x[:, k:k + 1].astype(np.float64) * W[:, k][None, :].astype(np.float64) + acc.astype(np.float64)).astype(F32)
    return acc if bias_first else (acc + b[None, :]).astype(F32)


def dev_order(K):
    return [8 * c + j for c in range((K + 7) // 8) for j in DEV8 if 8 * c + j < K]


def layer1_split(x, d_q, W, b, bias_first):
    """layer 1 as the device's separable form: Apre[t] (rollout features + bias) + Bpre[o] (obstacle features), two chains and one add.
    Feature order [x, sin x, cos x], x = [q (d_q), p (3)]."""
    d = x.shape[1] // 3
    qcols = [c for part in range(3) for c in range(part * d, part * d + d_q)]
    pcols = [c for part in range(3) for c in range(part * d + d_q, part * d + d)]
    a = chain(x[:, qcols], W[:, qcols], b, bias_first=bias_first)           # bias rides with the rollout half
    bb = chain(x[:, pcols], W[:, pcols], np.zeros_like(b), bias_first=True)
    return (a + bb).astype(F32)


def forward(m, x, variant):
    feats = orc.positional_encoding(x)
    n_q = x.shape[1] - 3
    zs, hs, cur = [], [feats], feats
    nl = len(m.W)
    for i in range(nl):
        W, b = m.W[i], m.b[i]
        if variant == "blas":
            z = (cur @ W.T + b).astype(F32)
        elif variant == "asc_bias_last":
            z = chain(cur, W, b)
        elif variant == "asc_bias_first":
            z = chain(cur, W, b, bias_first=i < nl - 1)                      # the device's last layer adds its bias afterwards
        elif variant == "dev8_bias_last":
            z = chain(cur, W, b, order=dev_order(cur.shape[1]))
        elif variant == "split_bias_last":
            z = layer1_split(cur, n_q, W, b, False) if i == 0 else chain(cur, W, b)
        elif variant == "split_bias_first":                                  # ~ the round-5 device
            z = layer1_split(cur, n_q, W, b, True) if i == 0 else chain(cur, W, b, order=dev_order(cur.shape[1]), bias_first=i < nl - 1)
        else:
            raise ValueError(variant)
        if i == nl - 1:
            return z, zs, hs
        zs.append(z)
        hs.append(orc._act(z, m.act))
        cur = np.concatenate((hs[-1], feats), axis=1) if i in m.skip_after else hs[-1]


def vjp(m, x, variant, bwd="blas"):
    """mlp_vjp_argmin with the forward in `variant`; the backward in BLAS order or as ascending chains (torch's mm)."""
    d = x.shape[1]
    y, zs, hs = forward(m, x, variant)
    feats = hs[0]
    mi = np.argmin(y, axis=1)
    g = m.W[-1][mi]
    g_feat = np.zeros_like(feats)
    for i in range(len(m.W) - 2, -1, -1):
        if i in m.skip_after:
            w = hs[i + 1].shape[1]
            g_feat = g_feat + g[:, w:]
            g = g[:, :w]
        gm = (g * orc._dact(zs[i], hs[i + 1], m.act)).astype(F32)
        g = (gm @ m.W[i]).astype(F32) if bwd == "blas" else chain(gm, np.ascontiguousarray(m.W[i].T), np.zeros(m.W[i].shape[1], F32), bias_first=True)
    g = g + g_feat
    grad = g[:, :d] + g[:, d:2 * d] * np.cos(x) - g[:, 2 * d:] * np.sin(x)
    return y, grad.astype(F32), mi


def step_velocity(m, q, fx, prm, mu, sg, al, variant, bwd="blas"):
    k = int(fx["k"])
    obs = fx["obs"]
    N, n = q.shape
    _, mind = orc.pass1_mindist(m, q, obs, fx["ignored_links"])
    idx = np.argsort(mind, axis=1, kind="stable")[:, :k]
    x = np.concatenate((np.repeat(q, k, axis=0), obs[idx.reshape(-1), :3]), axis=1).astype(F32)
    y, grad, mi = vjp(m, x, variant, bwd)
    if m.out_channels == 9:
        y = y / F32(100)
    y = y - obs[idx.reshape(-1), 3:4]
    dsel = y[np.arange(y.shape[0]), mi].reshape(N, k)
    g = grad[:, :n].reshape(N, k, n)
    e = np.exp(F32(-10) * dsel - (F32(-10) * dsel).max(axis=1, keepdims=True))
    w = (e / e.sum(axis=1, keepdims=True)).astype(F32)
    g_raw = (g * w[:, :, None]).sum(axis=1).astype(F32)
    return orc.modulation_step(q, fx["qf"], dsel[:, 0].copy(), g_raw, mu, sg, al, prm)["u"]


VARIANTS = [("blas", "blas", "numpy BLAS = the committed oracle"),
            ("asc_bias_last", "blas", "ascending-k fmaf chain from 0, bias last (= torch addmm)"),
            ("asc_bias_last", "asc", "  ... and the backward as ascending chains too (= torch mm)"),
            ("dev8_bias_last", "blas", "bias last, k order 8c + {0,4,1,5,2,6,3,7} (round-5 gemm256 order)"),
            ("asc_bias_first", "blas", "ascending chain, BIAS FIRST (accumulator init)"),
            ("split_bias_last", "blas", "bias last, layer 1 split into rollout half + obstacle half"),
            ("split_bias_first", "blas", "bias first + split layer 1 + dev8 order (~ the round-5 device)")]


def study(names):
    out = {}
    for name in names:
        fx = load(name)
        m = orc.Mlp.from_npz(weights_path(str(fx["kind"])))
        H, N = int(fx["H"]), int(fx["N"])
        dt = F32(fx["dt"])
        if float(dt) < 0.1:
            continue
        prm = orc.Params(dst_thr=float(fx["dst_thr"]), lin_thr=float(fx["lin_thr"]), p=int(fx["p"]), seds=seds_of(fx))
        for it in range(int(fx["n_iter"])):
            pre = f"it{it}_"
            ref = fx[pre + "all_traj"]
            mu, sg, al = fx[pre + "mu_tmp"], fx[pre + "sigma_tmp"], fx[pre + "alpha_tmp"]
            for i in range(1, H):
                q = np.ascontiguousarray(ref[:, i - 1, :])
                u_ref = (ref[:, i, :] - q) / dt
                for v, bwd, _ in VARIANTS:
                    c, _ = plain_bar(step_velocity(m, q, fx, prm, mu, sg, al, v, bwd), u_ref)
                    a = out.setdefault((name, v, bwd), dict(rows=0, plain=0, worst=0.0))
                    a["rows"] += c["rows"]; a["plain"] += c["plain"]; a["worst"] = max(a["worst"], c["worst"])
    return out


def torch_section():
    """torch.nn.functional.linear on CPU vs the ascending chain, bit for bit, per batch size M."""
    import torch
    m = orc.Mlp.from_npz(weights_path("franka"))
    rng = np.random.default_rng(0)
    print("torch", torch.__version__, "linear == ascending-k fmaf chain from 0 + bias (share of identical bits), Franka layer 2 (K = 256) / layer 1 (K = 30)")
    for M in (1, 5, 10, 11, 16, 64, 320, 20480):
        x = rng.uniform(-2, 2, (M, 10)).astype(F32)
        f = orc.positional_encoding(x)
        l1 = torch.nn.functional.linear(torch.from_numpy(f), torch.from_numpy(m.W[0]), torch.from_numpy(m.b[0])).numpy()
        h = np.maximum(l1, 0)
        l2 = torch.nn.functional.linear(torch.from_numpy(h), torch.from_numpy(m.W[1]), torch.from_numpy(m.b[1])).numpy()
        g = (torch.from_numpy(h) @ torch.from_numpy(m.W[1])).numpy()          # the vjp's product, W un-transposed
        gc = chain(h, np.ascontiguousarray(m.W[1].T), np.zeros(256, F32), bias_first=True)
        print(f"  M = {M:6d}: layer 1 {np.mean(l1 == chain(f, m.W[0], m.b[0])):.5f}   layer 2 {np.mean(l2 == chain(h, m.W[1], m.b[1])):.5f}"
              f"   backward product {np.mean(g == gc):.5f}")


if __name__ == "__main__":
    if "--torch" in sys.argv:
        torch_section()
    fams = {}
    names = [s for s in SCENARIOS if str(load(s)["kind"]) in ("franka", "planar7", "planar2")]
    res = study(names)
    for (name, v, bwd), a in res.items():
        f = fams.setdefault((name.split("_")[0], v, bwd), dict(rows=0, plain=0, worst=0.0))
        f["rows"] += a["rows"]; f["plain"] += a["plain"]; f["worst"] = max(f["worst"], a["worst"])
    print(f"{'pass-2 forward arithmetic / backward':88s}" + "".join(f"{fam:>24s}" for fam in ("franka", "planar2", "planar7")))
    for v, bwd, label in VARIANTS:
        row = f"{label:88s}"
        for fam in ("franka", "planar2", "planar7"):
            a = fams.get((fam, v, bwd))
            row += f"{100.0 * a['plain'] / a['rows']:9.3f} % / {a['worst']:8.1e}  " if a else " " * 24
        print(row)
    print("rows:", {fam: fams[(fam, 'blas', 'blas')]["rows"] for fam in ("franka", "planar2", "planar7") if (fam, 'blas', 'blas') in fams})

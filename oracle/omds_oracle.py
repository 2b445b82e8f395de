"""CPU ORACLE for the MPPI rollout + DS-modulation hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain numpy (float32) restatement of the reference algorithm, in the same
*unfused* op sequence as the reference (materialised [N*O, n+4] network input, dense layers,
sort / top-k, forward+backward on N*k rows, per-sample modulation, Euler step, cost, weighted
update).  It is the checker the HIP path is compared against; it is never the product path:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it (the package ``optimalmodulationds_amd`` never does, and fails loudly without its HIP
library).

Parity pinning: the reference ships NO tests, golden vectors or known-answer fixtures for this
path, so this oracle is pinned against outputs of the reference itself,
captured in this build container by ``tools/make_golden.py`` (which imports the reference from
/root/reference) and committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
every fixture (<= 1e-5 rel on trajectories / velocities, exact on indices).

Each function cites the reference file:line it follows (paths relative to
``/root/reference/python_scripts``; ``FN`` = ``ds_mppi/functions``, ``ML`` = ``mlp_learn/sdf``).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import chain

F32 = np.float32
FLT_MAX = np.finfo(np.float32).max


# ---------------------------------------------------------------------------------------------
# distance network: NeRF-encoded MLP, forward and vjp of the arg-min output
# ---------------------------------------------------------------------------------------------
@dataclass
class Mlp:
    """Weights ``W[i]`` are [out, in] like torch ``nn.Linear`` (ML/network_macros_mod.py:69-93)."""
    W: list
    b: list
    act: str = "relu"
    skip_after: tuple = ()   # indices of the Linear layers behind whose activations the encoded input is concatenated
                             # (MLPRegression skips, ML/network_macros_mod.py:113-146: y = layer(cat(y, x_nerf)))

    @property
    def out_channels(self):
        return self.W[-1].shape[0]

    @staticmethod
    def from_npz(path):
        z = np.load(path)
        n = len([k for k in z.files if k.startswith("W")])
        act = str(z["act"]) if "act" in z.files else "relu"
        skip = tuple(int(v) for v in z["skip_after"]) if "skip_after" in z.files else ()
        return Mlp([z[f"W{i}"].astype(F32) for i in range(n)], [z[f"b{i}"].astype(F32) for i in range(n)], act, skip)


def _act(z, act):
    return np.maximum(z, F32(0)) if act == "relu" else np.tanh(z)


def _dact(z, h, act):
    """Derivative of the activation given pre-activation z and output h."""
    return (z > 0).astype(F32) if act == "relu" else (F32(1) - h * h).astype(F32)


def positional_encoding(x):
    """x -> [x, sin x, cos x]  (ML/network_macros_mod.py:139-140), sin / cos as oracle/chain_arith.c restates them."""
    x = np.asarray(x, dtype=F32)
    return np.concatenate((x, chain.sin(x), chain.cos(x)), axis=-1).astype(F32)


def linear(h, W, b):
    """nn.Linear as torch-CPU computes it: ascending-k fmaf chains from zero, bias last (oracle/chain_arith.c)."""
    return chain.linear(h, W, b)


def pe_chain_rule(g, x):
    """Backward of x -> cat(x, sin x, cos x) as torch's autograd accumulates it: (g_x + g_cos * (-sin x)) + g_sin * cos x, every
    operation rounded on its own (bit for bit against the reference's gradients with torch's own sin / cos: tools/studies/
    assoc_order_study.py)."""
    x = np.asarray(x, dtype=F32)
    d = x.shape[1]
    g = np.asarray(g, dtype=F32)
    a = g[:, :d]
    b = (g[:, d:2 * d] * chain.cos(x)).astype(F32)
    c = (g[:, 2 * d:3 * d] * (-chain.sin(x))).astype(F32)
    return ((a + c).astype(F32) + b).astype(F32)


def mlp_forward(m: Mlp, x):
    """MLPRegression.forward (ML/network_macros_mod.py:137-146); with skips the encoded input is concatenated behind
    the output of each module (``torch.cat((y, x_nerf), dim=1)``)."""
    feats = positional_encoding(np.asarray(x, dtype=F32))
    h = feats
    for i in range(len(m.W) - 1):
        h = _act(linear(h, m.W[i], m.b[i]), m.act)
        if i in m.skip_after:
            h = np.concatenate((h, feats), axis=1)
    return linear(h, m.W[-1], m.b[-1])


def mlp_vjp_argmin(m: Mlp, x, seed=None):
    """functorch_vjp (ML/robot_sdf.py:153-158): forward, minIdx = argmin over ALL raw outputs,
    gradient of y[b, minIdx[b]] w.r.t. the n+3 inputs, as an analytic backward (masks + PE chain
    rule) instead of autograd.  ``seed`` [B]: differentiate those output columns instead (mlp_jacobian)."""
    x = np.asarray(x, dtype=F32)
    d = x.shape[1]
    feats = positional_encoding(x)
    hs, zs, cur = [feats], [], feats
    for i in range(len(m.W) - 1):
        z = linear(cur, m.W[i], m.b[i])
        zs.append(z)
        hs.append(_act(z, m.act))
        cur = np.concatenate((hs[-1], feats), axis=1) if i in m.skip_after else hs[-1]
    y = linear(cur, m.W[-1], m.b[-1])
    min_idx = np.argmin(y, axis=1) if seed is None else np.asarray(seed, dtype=np.int64)
    g = m.W[-1][min_idx]                                     # dy/d(input of the last layer)  [B, width]
    g_feat = np.zeros_like(feats)                            # direct paths into the encoded input (skip concatenations)
    for i in range(len(m.W) - 2, -1, -1):
        if i in m.skip_after:                                # g is w.r.t. cat(h_i, feats): split it
            w = hs[i + 1].shape[1]
            g_feat = g_feat + g[:, w:]
            g = g[:, :w]
        g = chain.matmul((g * _dact(zs[i], hs[i + 1], m.act)).astype(F32), m.W[i])    # -> grad wrt layer i input
    g = g + g_feat
    grad = pe_chain_rule(g, x)
    return y, grad, min_idx


def mlp_jacobian(m: Mlp, x, cols, order=None):
    """compute_signed_distance_wgrad with a column list (ML/robot_sdf.py:88-100): dist = y[:, order], grads[:, :, k] = gradient of
    dist[:, cols[k]] -- one backward per listed column of the RE-ORDERED outputs.  -> (dist [B,C'], grads [B,n+3,len(cols)])."""
    x = np.asarray(x, dtype=F32)
    order = list(range(m.W[-1].shape[0])) if order is None else list(order)
    y = mlp_forward(m, x)
    J = np.stack([mlp_vjp_argmin(m, x, seed=np.full(x.shape[0], order[c]))[1] for c in cols], axis=2)
    return y[:, order], J.astype(F32)


def mlp_closest_wgrad(m: Mlp, x, order=None):
    """compute_signed_distance_wgrad(q, 'closest') (ML/robot_sdf.py:101-109): arg-min over the re-ordered outputs, the gradient of
    that output.  -> (dist [B,C'], grads [B,n+3,1], minidx [B])."""
    x = np.asarray(x, dtype=F32)
    order = list(range(m.W[-1].shape[0])) if order is None else list(order)
    dist = mlp_forward(m, x)[:, order]
    mi = np.argmin(dist, axis=1)
    g = mlp_vjp_argmin(m, x, seed=np.asarray(order)[mi])[1]
    return dist, g[:, :, None], mi


# ---------------------------------------------------------------------------------------------
# distance + joint-space gradient over rollouts x obstacles
# ---------------------------------------------------------------------------------------------
def build_nn_input(q, obs):
    """Obstacle-major Cartesian product, row o*N + t = [q_t, obs_o(x,y,z,r)]  (FN/MPPI.py:93-95)."""
    q = np.asarray(q, dtype=F32)
    obs = np.asarray(obs, dtype=F32)
    return np.hstack((np.tile(q, (obs.shape[0], 1)), np.repeat(obs, q.shape[0], axis=0))).astype(F32)


def pass1_mindist(m: Mlp, q, obs, ignored_links):
    """First forward pass and per-(rollout, obstacle) min link distance  (FN/MPPI.py:233-243)."""
    n_in = q.shape[0]
    nn_input = build_nn_input(q, obs)
    nn_dist = mlp_forward(m, nn_input[:, :-1])
    if m.out_channels == 9:
        nn_dist = nn_dist / F32(100)
    nn_dist = nn_dist - nn_input[:, -1:]
    if len(ignored_links):
        nn_dist[:, list(ignored_links)] = F32(1e6)
    mindist = nn_dist.min(axis=1)
    return nn_input, mindist.reshape(obs.shape[0], n_in).T.copy()


def distance_repulsion_nn(m: Mlp, q, obs, k, ignored_links, softmax_k=-10.0):
    """MPPI.distance_repulsion_nn (FN/MPPI.py:227-282): returns (distance [N], nn_grad [N, n]) and
    the intermediates (min-distance matrix, sorted obstacle indices).  softmax_k is the reference's hard-coded -10
    (:277), a parameter here like in the C-ABI's omds_params."""
    q = np.asarray(q, dtype=F32)
    n_in, n_dof = q.shape
    nn_input, mind = pass1_mindist(m, q, obs, ignored_links)
    sort_idx = np.argsort(mind, axis=1, kind="stable")[:, :k]                 # :245-247
    rows = (np.arange(n_in)[:, None] + sort_idx * n_in).reshape(-1)           # :251
    nn_in2 = nn_input[rows]
    y, grad, min_idx = mlp_vjp_argmin(m, nn_in2[:, :-1])                      # :259
    if m.out_channels == 9:
        y = y / F32(100)
    y = y - nn_in2[:, -1:]
    d = y[np.arange(y.shape[0]), min_idx].reshape(n_in, k)                    # :270-272
    g = grad[:, :n_dof].reshape(n_in, k, n_dof)
    e = np.exp((F32(softmax_k) * d) - (F32(softmax_k) * d).max(axis=1, keepdims=True))    # softmax(-10 d)  :277
    w = (e / e.sum(axis=1, keepdims=True)).astype(F32)
    nn_grad = (g * w[:, :, None]).sum(axis=1).astype(F32)                     # :278
    return d[:, 0].copy(), nn_grad, mind, sort_idx


# ---------------------------------------------------------------------------------------------
# nominal DS, sigmoid, RBF policy
# ---------------------------------------------------------------------------------------------
def lin_ds_velocity(x, q_goal, lin_thr=0.015):
    """LinDS.get_velocity (FN/LinDS.py:11-21)."""
    x_dif = (x - q_goal).astype(F32)
    dst = np.sqrt((x_dif * x_dif).sum(axis=-1))
    y = (-x_dif).astype(F32)
    far = dst > F32(lin_thr)
    if far.any():
        y[far] = y[far] / dst[far][:, None]
    return y


def seds_velocity(x, q_goal, mu_in, b, sigma_inv, A, prior, den, lin_thr=1e-2, seds_thr=1e-2):
    """SEDS.get_velocity (FN/SEDS.py:34-74), per state: Gaussian mixture regression on x - q_goal with the per-component
    quantities SEDS.__init__ / GMR derive (mu_in, b = Mu halves; sigma_inv = inverse(Sigma_ii); A = Sigma_oi @ sigma_inv;
    den = sqrt(2 pi^n |det Sigma_ii| + 1e-100)); beta = clamp(nan_to_num(prior N / sum), 1e-8); farther than lin_thr from
    the goal the output is normalised, or replaced by the normalised linear DS where the mixture's output is below seds_thr.
    (The reference's own normalisation lines only broadcast for one state at a time; this is that per-state behaviour.)"""
    x = np.asarray(x, dtype=F32)
    xd = (x - np.asarray(q_goal, dtype=F32).reshape(1, -1)).astype(F32)
    dd = (xd[:, None, :] - mu_in[None]).astype(F32)                               # [N, G, n]
    prob = np.einsum("ngr,grc,ngc->ng", dd, sigma_inv, dd).astype(F32)
    with np.errstate(under="ignore", invalid="ignore", divide="ignore"):
        pxi = (prior[None] * (np.exp(F32(-0.5) * prob).astype(F32) / den[None])).astype(F32)
        beta = nan_to_num((pxi / pxi.sum(axis=1, keepdims=True, dtype=F32)).astype(F32))
    beta = np.maximum(beta, F32(1e-8))
    y = (beta[:, :, None] * (b[None] + np.einsum("grc,ngc->ngr", A, dd))).sum(axis=1).astype(F32)
    dst = np.linalg.norm(xd, axis=1).astype(F32)
    yn = np.linalg.norm(y, axis=1).astype(F32)
    far, weak = dst > F32(lin_thr), yn < F32(seds_thr)
    out = y.copy()
    with np.errstate(invalid="ignore", divide="ignore"):
        out[far] = y[far] / yn[far, None]
        lin = far & weak
        out[lin] = -xd[lin] / dst[lin, None]
    return out.astype(F32)


def generalized_sigmoid(x, y_min, y_max, x0, x1, k):
    """FN/MPPI.py:352-353."""
    with np.errstate(over="ignore"):
        return (F32(y_min) + F32(y_max - y_min) / (F32(1) + np.exp(F32(k) * (-x + F32((x0 + x1) / 2))))).astype(F32)


def eval_rbf(q, mu, sigma, p=2):
    """FN/policy.py:186-199: phi[t, kappa] = exp(-sigma * ||q - mu||_p^2)."""
    diff = np.abs(q[:, None, :] - mu)
    if p == 2:
        nrm = np.sqrt((diff * diff).sum(axis=2))
    else:
        nrm = (diff ** F32(p)).sum(axis=2) ** F32(1.0 / p)
    return np.exp(-sigma * nrm * nrm).astype(F32)


def nan_to_num(x):
    return np.nan_to_num(x, nan=0.0, posinf=FLT_MAX, neginf=-FLT_MAX).astype(F32)


def qr_basis(nn_grad):
    """E = qr([g, e_2 .. e_n]).Q with column 0 overwritten by +g/||g||  (FN/MPPI.py:122-127)."""
    N, n = nn_grad.shape
    E = np.empty((N, n, n), dtype=F32)
    for t in range(N):
        A = np.eye(n, dtype=F32)
        A[:, 0] = nn_grad[t]
        Q, _ = np.linalg.qr(A)
        E[t] = Q
    E[:, :, 0] = nn_grad / np.sqrt((nn_grad * nn_grad).sum(axis=1))[:, None]
    return E


# ---------------------------------------------------------------------------------------------
# the rollout loop
# ---------------------------------------------------------------------------------------------
@dataclass
class Params:
    """Hard-coded constants of FN/MPPI.py:132-155,193-216 and LinDS, as parameters."""
    dst_thr: float = 0.5
    lin_thr: float = 0.015
    p: int = 2
    lvel: tuple = (0.0, 1.0, -1.0, 0.0, 10.0)          # :132
    ln: tuple = (0.0, 1.0, 0.0, 0.1, 100.0)            # :149-153
    ltau: tuple = (5.0, 1.0, 0.0, 0.1, 100.0)          # :155
    goal_act_cut: float = 0.5                          # :194
    norm_clamp: float = 0.5                            # :212
    coll_slow: float = 0.1                             # :215
    coll_repulse: float = 0.1                          # :216
    softmax_k: float = -10.0                           # :277
    want_basis: bool = False
    # FN/MPPI_toy.py variant: nominal DS (q - qf) @ A (:89) and kernel values stored times activation (:178-179)
    A: object = None
    kval_times_act: bool = False
    # SEDS nominal DS (FN/SEDS.py): dict(mu_in, b, sigma_inv, A, prior, den, lin_thr, seds_thr) as seds_velocity takes them
    seds: object = None


@dataclass
class RolloutOut:
    all_traj: np.ndarray
    closest_dist_all: np.ndarray
    kernel_val_all: np.ndarray
    dot_products: np.ndarray
    kernel_activations: np.ndarray
    qdot: np.ndarray
    norm_basis_n: np.ndarray
    norm_basis: np.ndarray | None = None
    extras: dict = field(default_factory=dict)


def modulation_step(q_prev, qf, distance_raw, g_raw, mu_tmp, sigma_tmp, alpha_tmp, prm: Params = Params()):
    """Everything of one horizon step AFTER the distance network (FN/MPPI.py:102-217): nominal DS,
    eigenvalues, RBF policy, activations, closed-form M v, normalisation, collision handling.
    ``distance_raw`` is the network distance of the closest obstacle (before ``dst_thr``),
    ``g_raw`` the blended joint-space gradient.  Returns a dict of per-rollout results."""
    q_prev = np.asarray(q_prev, dtype=F32)
    K = mu_tmp.shape[1]
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        if prm.seds is not None:
            v = seds_velocity(q_prev, qf, **prm.seds)                             # :106 with DS = SEDS
        elif prm.A is None:
            v = lin_ds_velocity(q_prev, qf, prm.lin_thr)                          # :106
        else:
            v = ((q_prev - qf).astype(F32) @ np.asarray(prm.A, dtype=F32)).astype(F32)   # MPPI_toy.py:89
        vnorm = np.sqrt((v * v).sum(axis=1)).reshape(-1, 1)
        vhat = v / vnorm
        distance = (distance_raw - F32(prm.dst_thr)).astype(F32)                   # :117
        ghat = (g_raw / np.sqrt((g_raw * g_raw).sum(axis=1))[:, None]).astype(F32)  # :126
        dot = (ghat * vhat).sum(axis=-1)                                           # :129
        l_vel = generalized_sigmoid(dot, *prm.lvel)                                # :132
        l_n = generalized_sigmoid(distance, *prm.ln)
        l_nv = l_vel * F32(1) + (F32(1) - l_vel) * l_n                              # :154
        l_tau = generalized_sigmoid(distance, *prm.ltau)
        if K > 0:                                                                  # :165-186
            phi = eval_rbf(q_prev, mu_tmp, sigma_tmp, prm.p)
            pol = (alpha_tmp * phi[:, :, None]).sum(axis=1).astype(F32)
        else:
            phi = np.zeros((q_prev.shape[0], 0), dtype=F32)
            pol = v * F32(0)
        ca = (F32(1) - l_n)[:, None]
        va = (F32(1) - l_vel)[:, None]
        ga = (np.sqrt(np.abs(q_prev - qf)).sum(axis=1) ** 2).clip(0, 1)[:, None].astype(F32)  # norm(p=0.5)
        ga[ga < F32(prm.goal_act_cut)] = 0
        act = ca * va * ga
        v_tot = v + act * pol * vnorm                                              # :197-206
        # M v with M = E diag(l_nv, l_tau, ...) E^T  ==  l_tau v + (l_nv - l_tau)(g.v) g   (:161,209)
        u = l_tau[:, None] * v_tot + ((l_nv - l_tau) * (ghat * v_tot).sum(axis=1))[:, None] * ghat
        unorm = np.sqrt((u * u).sum(axis=1)).reshape(-1, 1)
        s = unorm.copy()
        s[s <= F32(prm.norm_clamp)] = 1                                            # :212
        u = nan_to_num(u / s)
        coll = distance < 0
        u[coll] *= F32(prm.coll_slow)                                              # :215
        rep = ghat * vnorm * F32(prm.coll_repulse)
        u[coll] += rep[coll]                                                       # :217
    return dict(u=u.astype(F32), distance=distance, ghat=ghat, dot=dot.astype(F32), act=act[:, 0].astype(F32),
                phi=phi, unorm=unorm[:, 0], ga=ga[:, 0])


def propagate(m: Mlp, q_cur, qf, obs, *, N, H, dt, k, ignored_links, mu_tmp, sigma_tmp, alpha_tmp,
              prm: Params = Params()):
    """MPPI.propagate (FN/MPPI.py:97-224).  ``mu_tmp [N,K,n]``, ``sigma_tmp [N,K]``,
    ``alpha_tmp [N,K,n]`` are the sampled policy tensors (first K kernels).  ``q_cur`` is [n]
    (the reference) or [N, n] (per-rollout starts, used by teacher-forced tests)."""
    q_cur = np.asarray(q_cur, dtype=F32)
    qf = np.asarray(qf, dtype=F32)
    n = q_cur.shape[-1]
    K = mu_tmp.shape[1]
    dt = F32(dt)
    all_traj = np.zeros((N, H, n), dtype=F32)
    dist_all = np.full((N, H), 100, dtype=F32)
    kval_all = np.zeros((N, H, K), dtype=F32)
    dots = np.zeros((N, H), dtype=F32)
    acts = np.zeros((N, H), dtype=F32)
    nb_n = np.zeros((N, H, n), dtype=F32)
    nb = np.zeros((N, H, n, n), dtype=F32) if prm.want_basis else None
    qdot = np.zeros((N, n), dtype=F32)
    all_traj[:, 0, :] = q_cur
    for i in range(1, H + 1):
        q_prev = all_traj[:, i - 1, :]
        d_raw, g_raw, _, _ = distance_repulsion_nn(m, q_prev, obs, k, ignored_links, prm.softmax_k)   # :113
        st = modulation_step(q_prev, qf, d_raw, g_raw, mu_tmp, sigma_tmp, alpha_tmp, prm)
        dist_all[:, i - 1] = st["distance"]
        nb_n[:, i - 1] = st["ghat"]
        if nb is not None:
            with np.errstate(invalid="ignore", divide="ignore"):
                nb[:, i - 1] = qr_basis(g_raw)
        dots[:, i - 1] = st["dot"]
        acts[:, i - 1] = st["act"]
        if K > 0:
            kval_all[:, i - 1, :] = st["phi"] * st["act"][:, None] if prm.kval_times_act else st["phi"]
        if i < H:
            all_traj[:, i, :] = all_traj[:, i - 1, :] + dt * st["u"]              # :221
        if i == 1:
            qdot = st["u"].copy()
    return RolloutOut(all_traj, dist_all, kval_all, dots, acts, qdot, nb_n, nb)


def relu_margin(m: Mlp, x):
    """Per row: min over hidden layers of  min_j |z_j| / max_j |z_j|  (z = pre-activations).
    fp32 accumulation-order differences perturb z by ~1e-6 of the layer's scale, so a row whose
    margin is below a few 1e-6 may flip a ReLU mask there and change the vjp discretely (SURVEY
    section 7 hard part b); parity tests use this to flag such rows."""
    h = positional_encoding(np.asarray(x, dtype=F32))
    marg = np.full(h.shape[0], np.inf, dtype=F32)
    if m.act != "relu":
        return marg                      # smooth activation: no masks to flip
    feats = h
    for i in range(len(m.W) - 1):
        z = linear(h, m.W[i], m.b[i])
        az = np.abs(z)
        marg = np.minimum(marg, az.min(axis=1) / np.maximum(az.max(axis=1), F32(1e-30)))
        h = _act(z, m.act)
        if i in m.skip_after:
            h = np.concatenate((h, feats), axis=1)
    return marg


def rollout_relu_margin(m: Mlp, q, obs, sort_idx):
    """relu_margin of the k selected (rollout, obstacle) rows, min over k -> one value per rollout."""
    q = np.asarray(q, dtype=F32)
    N, k = sort_idx.shape
    x = np.concatenate((np.repeat(q, k, axis=0), np.asarray(obs, dtype=F32)[sort_idx.reshape(-1), :3]), axis=1)
    return relu_margin(m, x).reshape(N, k).min(axis=1)


def blended_gradient_alternatives(m: Mlp, q_row, obs, idx_row, margin, softmax_k=-10.0, max_units=10):
    """All blended gradients (FN/MPPI.py:270-278) one rollout can legitimately have under fp32 rounding of the forward pass.

    The vjp of a ReLU network multiplies by the masks 1[z > 0]; a hidden pre-activation within ``margin`` (relative to the
    layer's largest |z|) of zero may come out on either side of it depending on the summation order of the dot product --
    the reference's BLAS, the numpy oracle and an MFMA chain all round differently -- and each choice is a valid fp32
    evaluation of the same network.  This enumerates every assignment of those ambiguous units over the rollout's k rows
    (``idx_row`` = the k closest obstacles, ascending) and returns the list of resulting blended gradients [n_alt, n];
    entry 0 is the oracle's own.  Rows with more than ``max_units`` ambiguous units in total return only entry 0."""
    q_row = np.asarray(q_row, dtype=F32).reshape(1, -1)
    n_dof = q_row.shape[1]
    k = len(idx_row)
    x = np.concatenate((np.repeat(q_row, k, axis=0), np.asarray(obs, dtype=F32)[np.asarray(idx_row), :3]), axis=1)
    d = x.shape[1]
    feats = positional_encoding(x)
    hs, zs, cur = [feats], [], feats
    for i in range(len(m.W) - 1):
        z = linear(cur, m.W[i], m.b[i])
        zs.append(z)
        hs.append(_act(z, m.act))
        cur = np.concatenate((hs[-1], feats), axis=1) if i in m.skip_after else hs[-1]
    y = linear(cur, m.W[-1], m.b[-1])
    min_idx = np.argmin(y, axis=1)
    yd = y / F32(100) if m.out_channels == 9 else y
    dist = (yd - np.asarray(obs, dtype=F32)[np.asarray(idx_row), 3:4])[np.arange(k), min_idx]
    e = np.exp(F32(softmax_k) * dist - (F32(softmax_k) * dist).max())
    w = (e / e.sum()).astype(F32)
    amb = []                                     # (row, layer, unit)
    if m.act == "relu":
        for i, z in enumerate(zs):
            az = np.abs(z)
            r, j = np.nonzero(az < F32(margin) * az.max(axis=1, keepdims=True))
            amb += [(int(a), i, int(b)) for a, b in zip(r, j)]
    masks0 = [_dact(zs[i], hs[i + 1], m.act) for i in range(len(zs))]

    def grad_with(masks):
        g = m.W[-1][min_idx]
        g_feat = np.zeros_like(feats)
        for i in range(len(m.W) - 2, -1, -1):
            if i in m.skip_after:
                g_feat = g_feat + g[:, hs[i + 1].shape[1]:]
                g = g[:, :hs[i + 1].shape[1]]
            g = chain.matmul((g * masks[i]).astype(F32), m.W[i])
        g = g + g_feat
        gx = pe_chain_rule(g, x)
        return (gx[:, :n_dof].astype(F32) * w[:, None]).sum(axis=0).astype(F32)

    out = [grad_with(masks0)]
    if 0 < len(amb) <= max_units:
        for code in range(1, 1 << len(amb)):
            masks = [mk.copy() for mk in masks0]
            for bit, (r, i, j) in enumerate(amb):
                if (code >> bit) & 1:
                    masks[i][r, j] = F32(1) - masks[i][r, j]
            out.append(grad_with(masks))
    return np.stack(out)


# ---------------------------------------------------------------------------------------------
# cost (FN/cost.py) and forward kinematics (FN/fk_num.py)
# ---------------------------------------------------------------------------------------------
def dh_transform(q, d, theta, a, alpha):
    """Modified-DH link transform (FN/fk_num.py:7-27)."""
    sa, ca = np.sin(F32(alpha)), np.cos(F32(alpha))
    sq, cq = np.sin(F32(q + theta)), np.cos(F32(q + theta))
    return np.array([[cq, -sq, 0.0, a],
                     [sq * ca, cq * ca, -sa, -d * sa],
                     [sq * sa, cq * sa, ca, d * ca],
                     [0.0, 0.0, 0.0, 1.0]], dtype=F32)


def link_endpoints(q, dh_params):
    """Last sample point of every link: frame_{i+1} applied to [a_{i+1}, 0, 0]
    (FN/fk_num.py:30-75 with n_pts=2, the [:, :, -1, :] slice of FN/cost.py:28)."""
    n = len(q)
    T = np.eye(4, dtype=F32)
    pts = np.zeros((n, 3), dtype=F32)
    for i in range(n):
        d, theta, a, alpha = dh_params[i]
        T = (T @ dh_transform(q[i], d, theta, a, alpha)).astype(F32)
        p1 = np.array([dh_params[i + 1, 2], 0, 0], dtype=F32)
        pts[i] = T[:3, :3] @ p1 + T[:3, 3]
    return pts


def evaluate_costs(all_traj, closest_dist_all, qf, dh_params, q_min, q_max, terms=("goal", "coll", "jl", "stag", "fk")):
    """Cost.evaluate_costs (FN/cost.py:13-46); FN/cost_toy.py:14-18 sums ("goal", "coll", "stag") only."""
    q_end = all_traj[:, -1, :]
    goal = F32(10) * np.sqrt(((q_end - qf) ** 2).sum(axis=1))
    coll = F32(100) * (closest_dist_all < 0).sum(axis=1)
    viol = ((all_traj < q_min).sum(axis=1) + (all_traj > q_max).sum(axis=1)).sum(axis=1)
    jl = F32(100) * (viol > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        dist = np.sqrt(((all_traj[:, 0, :] - q_end) ** 2).sum(axis=1))
        stag = F32(10) * goal * nan_to_num(F32(1) / dist)
    goal_fk = link_endpoints(qf, dh_params)
    fk = np.zeros(all_traj.shape[0], dtype=F32)
    for t in range(all_traj.shape[0]):
        diff = link_endpoints(q_end[t], dh_params) - goal_fk
        fk[t] = np.sqrt((diff * diff).sum(axis=1)).sum()
    fk = F32(10) * fk
    parts = dict(goal=goal, coll=coll, jl=jl, stag=stag, fk=fk)
    total = np.zeros_like(goal)
    for name in ("goal", "coll", "jl", "stag", "fk"):          # left to right like the reference
        if name in terms:
            total = total + parts[name]
    return total.astype(F32), dict(goal=goal.astype(F32), coll=coll, jl=jl, stag=stag, fk=fk.astype(F32))


# ---------------------------------------------------------------------------------------------
# MPPI weights and policy update (FN/MPPI.py:319-345, FN/policy.py:88-113)
# ---------------------------------------------------------------------------------------------
def mppi_weights(cost):
    beta = cost.mean(dtype=F32) / F32(50)
    w = np.exp(F32(-1) / beta * cost).astype(F32)
    return (w / w.sum(dtype=F32)).astype(F32)


def shift_policy_means(cost, kernel_val_all, kernel_activations, mu_c, sigma_c, alpha_c,
                       mu_tmp, sigma_tmp, alpha_tmp, ker_thr, rate, toy=False):
    """Returns new (mu_c, sigma_c, alpha_c), update mask, weights.  K = mu_c.shape[0].
    toy=True: FN/MPPI_toy.py:314-323 -- kernel_val_all already holds phi*activation and there is no
    rollout-0 mask."""
    w = mppi_weights(cost)
    K = mu_c.shape[0]
    if K == 0:
        return mu_c, sigma_c, alpha_c, np.zeros(0, dtype=bool), w
    with np.errstate(invalid="ignore"):
        prod = kernel_val_all if toy else kernel_val_all * kernel_activations[:, :, None]
        mx = np.where(np.isnan(prod).any(axis=1), np.nan, np.nanmax(np.where(np.isnan(prod), -np.inf, prod), axis=1))
        mask = mx.mean(axis=0) > F32(ker_thr)                                   # FN/MPPI.py:336-339
        mask_base = kernel_val_all[0].mean(axis=0) > F32(ker_thr)               # :341
    if not toy:
        mask = mask & mask_base
    upd = np.where(mask, F32(rate), F32(0)).astype(F32)
    mu_sum = (w[:, None, None] * mu_tmp).sum(axis=0)
    sg_sum = (w[:, None] * sigma_tmp).sum(axis=0)
    al_sum = (w[:, None, None] * alpha_tmp).sum(axis=0)
    mu_new = (F32(1) - upd[:, None]) * mu_c + upd[:, None] * mu_sum
    sg_new = (F32(1) - upd) * sigma_c + upd * sg_sum
    al_new = (F32(1) - upd[:, None]) * alpha_c + upd[:, None] * al_sum
    return mu_new.astype(F32), sg_new.astype(F32), al_new.astype(F32), mask, w


def get_qdot(cost, qdot, mode="best"):
    """MPPI.get_qdot (FN/MPPI.py:319-329)."""
    if mode == "best":
        return qdot[int(np.argmin(cost))]
    w = mppi_weights(cost)
    return (w[:, None] * qdot).sum(axis=0).astype(F32)


def check_traj_for_kernels(all_traj, dist_all, dots_all, mu_c, sigma_c, thr_dist, thr_kernel, thr_dot, p=2):
    """TensorPolicyMPPI.check_traj_for_kernels (FN/policy.py:153-175): candidate kernel centres."""
    sel = (dist_all < thr_dist) & (dots_all < thr_dot)
    cand = all_traj[sel].reshape(-1, all_traj.shape[-1])
    if mu_c.shape[0] > 0 and cand.shape[0] > 0:
        diff = np.abs(cand[:, None, :] - mu_c)
        nrm = np.sqrt((diff * diff).sum(axis=-1)) if p == 2 else (diff ** p).sum(axis=-1) ** (1.0 / p)
        rbf = np.exp(-sigma_c * nrm * nrm)
        cand = cand[rbf.max(axis=-1) < thr_kernel]
    return cand

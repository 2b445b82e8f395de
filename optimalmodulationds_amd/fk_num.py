"""Host-side link sample points -- mirrors ``ds_mppi/functions/fk_num.py`` (``numeric_fk_model`` :50-75,
``numeric_fk_model_vec`` :78-89, ``dh_fk`` :30-47).  The drivers use them for visualisation payloads only
(frankaPlanner.py:162-177: ``kernel_fk``, ``best_traj_fk``); the FK term of the cost runs on the device
(``k_cost``).  Batched over configurations instead of the reference's per-configuration TorchScript loop."""
from __future__ import annotations

import numpy as np
import torch


def _frames(q, dh_params) -> np.ndarray:
    """[B, n+1, 4, 4] float32: identity followed by the cumulative modified-DH transforms of joints 0..n-1 (numpy inside: a few
    dozen 4x4 products, where every torch op would cost more in dispatch than in arithmetic)."""
    q = np.asarray(q, dtype=np.float32)
    dh = np.asarray(dh_params, dtype=np.float32)
    B, n = q.shape
    d, theta, a, alpha = (dh[:n, c] for c in range(4))
    sa, ca = np.sin(alpha), np.cos(alpha)
    sq, cq = np.sin(q + theta), np.cos(q + theta)
    M = np.zeros((B, n, 4, 4), np.float32)
    M[:, :, 0, 0], M[:, :, 0, 1], M[:, :, 0, 3] = cq, -sq, a
    M[:, :, 1, 0], M[:, :, 1, 1], M[:, :, 1, 2], M[:, :, 1, 3] = sq * ca, cq * ca, -sa, -d * sa
    M[:, :, 2, 0], M[:, :, 2, 1], M[:, :, 2, 2], M[:, :, 2, 3] = sq * sa, cq * sa, ca, d * ca
    M[:, :, 3, 3] = 1.0
    T = np.empty((B, n + 1, 4, 4), np.float32)
    T[:, 0] = np.eye(4, dtype=np.float32)
    for i in range(n):
        T[:, i + 1] = T[:, i] @ M[:, i]
    return T


def numeric_fk_model_vec(q, dh_params, n_pts: int):
    """q [B, n] -> (link_pts [B, n, n_pts, 3] in the base frame, pts_int [B, n, n_pts, 3] in the link frames);
    link i is sampled at linspace(0.01, 1, n_pts) * [a_{i+1}, 0, 0] in frame i+1."""
    qn = torch.as_tensor(q, dtype=torch.float32).detach().cpu().numpy()
    dh = torch.as_tensor(dh_params, dtype=torch.float32).detach().cpu().numpy()
    B, n = qn.shape
    T = _frames(qn, dh)[:, 1:]                                  # [B, n, 4, 4]
    span = torch.linspace(0.01, 1, n_pts, dtype=torch.float32).numpy()   # [P], the reference's torch.linspace (its float32 values)
    local = np.zeros((n, n_pts, 3), np.float32)
    local[:, :, 0] = dh[1:n + 1, 2][:, None] * span[None, :]
    # R @ local + t as a broadcast multiply-sum (a batched GEMM through torch.einsum woke the whole intra-op thread pool for a few
    # dozen configurations; its workers then spin on all cores and starve the HIP runtime's helper threads -- measured in the
    # planner loop on a 128-core host: 8 ms for that line and omds_propagate 5.9 -> 22 ms per call)
    pts = (T[:, :, None, :3, :3] * local[None, :, :, None, :]).sum(-1) + T[:, :, None, :3, 3]
    return torch.from_numpy(pts.astype(np.float32)), torch.from_numpy(np.broadcast_to(local, (B, n, n_pts, 3)).copy())


def numeric_fk_model(q: torch.Tensor, dh_params: torch.Tensor, n_pts: int):
    """Single configuration q [n] -> (links [n, n_pts, 3], pts_int [n, n_pts, 3])."""
    pts, local = numeric_fk_model_vec(torch.as_tensor(q, dtype=torch.float32)[None], dh_params, n_pts)
    return pts[0], local[0]


def dh_fk(q: torch.Tensor, dh_params: torch.Tensor):
    """List of the n+1 cumulative 4x4 frames (identity first), like the reference's ``dh_fk``."""
    T = torch.from_numpy(_frames(torch.as_tensor(q, dtype=torch.float32).numpy()[None], torch.as_tensor(dh_params, dtype=torch.float32).numpy())[0])
    return [T[i] for i in range(T.shape[0])]

"""Host-side link sample points -- mirrors ``ds_mppi/functions/fk_num.py`` (``numeric_fk_model`` :50-75,
``numeric_fk_model_vec`` :78-89, ``dh_fk`` :30-47).  The drivers use them for visualisation payloads only
(frankaPlanner.py:162-177: ``kernel_fk``, ``best_traj_fk``); the FK term of the cost runs on the device
(``k_cost``).  Batched over configurations instead of the reference's per-configuration TorchScript loop."""
from __future__ import annotations

import torch


def _frames(q: torch.Tensor, dh_params: torch.Tensor) -> torch.Tensor:
    """[B, n+1, 4, 4]: identity followed by the cumulative modified-DH transforms of joints 0..n-1."""
    B, n = q.shape
    d, theta, a, alpha = (dh_params[:n, c].to(q.dtype) for c in range(4))
    sa, ca = torch.sin(alpha), torch.cos(alpha)
    sq, cq = torch.sin(q + theta), torch.cos(q + theta)
    zero, one = torch.zeros_like(sq), torch.ones_like(sq)
    rows = [torch.stack((cq, -sq, zero, a.expand(B, n)), -1),
            torch.stack((sq * ca, cq * ca, (-sa).expand(B, n), (-d * sa).expand(B, n)), -1),
            torch.stack((sq * sa, cq * sa, ca.expand(B, n), (d * ca).expand(B, n)), -1),
            torch.stack((zero, zero, zero, one), -1)]
    M = torch.stack(rows, -2)                                   # [B, n, 4, 4]
    T = [torch.eye(4, dtype=q.dtype).expand(B, 4, 4)]
    for i in range(n):
        T.append(T[-1] @ M[:, i])
    return torch.stack(T, 1)


def numeric_fk_model_vec(q: torch.Tensor, dh_params: torch.Tensor, n_pts: int):
    """q [B, n] -> (link_pts [B, n, n_pts, 3] in the base frame, pts_int [B, n, n_pts, 3] in the link frames);
    link i is sampled at linspace(0.01, 1, n_pts) * [a_{i+1}, 0, 0] in frame i+1."""
    q = torch.as_tensor(q, dtype=torch.float32)
    dh_params = torch.as_tensor(dh_params, dtype=torch.float32)
    B, n = q.shape
    T = _frames(q, dh_params)[:, 1:]                            # [B, n, 4, 4]
    span = torch.linspace(0.01, 1, n_pts, dtype=q.dtype)        # [P]
    local = torch.zeros(n, n_pts, 3, dtype=q.dtype)
    local[:, :, 0] = dh_params[1:n + 1, 2].to(q.dtype)[:, None] * span[None, :]
    pts = torch.einsum('bnij,npj->bnpi', T[:, :, :3, :3], local) + T[:, :, None, :3, 3]
    return pts, local.expand(B, n, n_pts, 3).clone()


def numeric_fk_model(q: torch.Tensor, dh_params: torch.Tensor, n_pts: int):
    """Single configuration q [n] -> (links [n, n_pts, 3], pts_int [n, n_pts, 3])."""
    pts, local = numeric_fk_model_vec(torch.as_tensor(q, dtype=torch.float32)[None], dh_params, n_pts)
    return pts[0], local[0]


def dh_fk(q: torch.Tensor, dh_params: torch.Tensor):
    """List of the n+1 cumulative 4x4 frames (identity first), like the reference's ``dh_fk``."""
    T = _frames(torch.as_tensor(q, dtype=torch.float32)[None], torch.as_tensor(dh_params, dtype=torch.float32))[0]
    return [T[i] for i in range(T.shape[0])]

"""The 2-D toy drivers' variant of the planner -- mirrors ``ds_mppi/functions/MPPI_toy.py`` (class MPPI) and
``cost_toy.py``: nominal DS ``(q - qf) @ A`` (MPPI_toy.py:89), its own constants (:114,126-133,176,199),
kernel values stored times the activation (:178-179), update mask without the rollout-0 term (:314-323),
cost = goal + collision + stagnation (cost_toy.py:14-18), ``propagate()`` returns a 4-tuple (:207).
Obstacles are planar discs [x, y, r] and the network takes DOF + 2 inputs (scripts/standaloneToy2d.py:33,58-66).
Same HIP path as ``mppi.MPPI``; the differences are parameters of the C-ABI (``omds_params.variant`` /
``cost_terms``, ``omds_set_ds_matrix``)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from .mppi import MPPI as _MPPI, _np


class _MatrixDS:
    """Stands in for the DS object of ``mppi.MPPI``: goal + matrix, no ``lin_thr`` normalisation."""
    lin_thr = 0.0

    def __init__(self, qf, A):
        self.q_goal = torch.as_tensor(_np(qf))
        self.A = torch.as_tensor(_np(A))

    def get_velocity(self, x):
        return (x - self.q_goal) @ self.A


def toy_params(p):
    """The constants MPPI_toy.py hard-codes, written into an ``omds_params`` struct."""
    p.lvel[:] = (0.0, 1.0, -0.2, 0.0, 100.0)       # :114
    p.ln[:] = (0.0, 1.0, 0.0, 0.5, 30.0)           # :124-131 (planar branch)
    p.ltau[:] = (3.0, 1.0, 0.0, 0.5, 30.0)         # :130,133 (y_min = ltau_max)
    p.goal_act_cut = 0.3                           # :176
    p.coll_repulse = 0.05                          # :199
    p.variant = L.VARIANT_KVAL_TIMES_ACT | L.VARIANT_NO_BASE_MASK
    p.cost_terms = L.COST_GOAL | L.COST_COLLISION | L.COST_STAGNATION
    return p


class MPPI(_MPPI):
    def __init__(self, q0, qf, dh_params, obs, dt, dt_H, N_traj, A, dh_a, nn_model, n_closest_obs, **kw):
        obs3 = _np(obs).reshape(-1, 3)
        obs4 = np.zeros((obs3.shape[0], 4), np.float32)
        obs4[:, :2] = obs3[:, :2]
        obs4[:, 3] = obs3[:, 2]
        self.A = torch.as_tensor(_np(A))
        n = _np(q0).reshape(-1).shape[0]
        dh_params = _np(dh_params).reshape(-1, 4)[:n + 1]       # the toy driver passes a dummy 4x4 (standaloneToy2d.py:56)
        super().__init__(q0, qf, dh_params, obs4, dt, dt_H, N_traj, [_MatrixDS(qf, A)], dh_a, nn_model, n_closest_obs, **kw)
        self.dst_thr = 0.1                          # MPPI_toy.py:56
        toy_params(self._engine.params)

    def update_obstacles(self, obs):
        obs3 = _np(obs).reshape(-1, 3)
        obs4 = np.zeros((obs3.shape[0], 4), np.float32)
        obs4[:, :2] = obs3[:, :2]
        obs4[:, 3] = obs3[:, 2]
        return super().update_obstacles(obs4)

    def _push(self):
        before = self._pushed_version
        super()._push()
        a = _np(self.A).tobytes()
        if self._pushed_version != before or a != getattr(self, "_pushed_A", None):   # the base class re-installed the linear DS, or A changed
            self._engine.set_ds_matrix(_np(self.qf), _np(self.A))
            self._pushed_A = a
            self._pushed_version = self._engine.config_version

    def propagate(self, fetch=None):
        r = super().propagate(fetch)
        return None if r is None else r[:4]

    def shift_policy_means(self):
        super().shift_policy_means()
        return 0

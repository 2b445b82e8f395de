"""Nominal linear DS -- mirrors ``ds_mppi/functions/LinDS.py`` (class LinDS, lines 6-21).

Inside ``MPPI.propagate`` the velocity is evaluated on the GPU (k_modulate); this class is the
parameter holder the reference's callers construct (``DS_ARRAY = [LinDS(q_f), LinDS(q_0)]``,
frankaPlanner.py:65-67)."""
import numpy as np
import torch


class LinDS:
    def __init__(self, q_goal):
        self.q_goal = torch.as_tensor(np.asarray(q_goal, dtype=np.float32))
        self.lin_thr = 0.015
        self.dof = self.q_goal.shape[0]

    def get_velocity(self, x):
        """Host convenience with the reference's semantics (LinDS.py:11-21); the rollouts never
        call it -- they use the device kernel."""
        x = torch.as_tensor(x, dtype=torch.float32)
        x_dif = x - self.q_goal
        dst = x_dif.norm(p=2, dim=-1)
        y = -x_dif
        far = dst > self.lin_thr
        if far.ndim == 0:
            return y / dst if bool(far) else y
        y[far] = y[far] / dst[far].unsqueeze(-1)
        return y

"""Cost parameters -- mirrors ``ds_mppi/functions/cost.py`` (class Cost, lines 4-46).

The evaluation runs on the GPU (k_cost, incl. the modified-DH forward kinematics); this object
holds what the reference's callers read and mutate (``mppi.Cost.q_min/q_max``,
standalonePlanar2d.py:128-129)."""
import numpy as np
import torch

FRANKA_Q_MIN = [-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973]   # cost.py:10
FRANKA_Q_MAX = [2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973]         # cost.py:11


class Cost:
    def __init__(self, q_f, dh_params, owner=None):
        self.qf = torch.as_tensor(np.asarray(q_f, dtype=np.float32))
        self.COLL_WEIGHT = 500
        self.dh_params = torch.as_tensor(np.asarray(dh_params, dtype=np.float32))
        n = self.qf.shape[0]
        # the reference always installs the 7-DoF Franka limits (and crashes for other robots
        # unless the driver overrides them); for n != 7 we start from +-inf-like wide limits
        if n == 7:
            self.q_min = torch.tensor(FRANKA_Q_MIN)
            self.q_max = torch.tensor(FRANKA_Q_MAX)
        else:
            self.q_min = torch.full((n,), -1e30)
            self.q_max = torch.full((n,), 1e30)
        self.rest = self.q_min + (self.q_max - self.q_min) * 0.5
        self._owner = owner

    def evaluate_costs(self, all_traj=None, closest_dist_all=None):
        """Cost of the owner's CURRENT rollouts (cost.py:13-22), evaluated on the device.  The
        tensor arguments are accepted for signature compatibility; the device copies are used."""
        if self._owner is None:
            raise RuntimeError("Cost.evaluate_costs needs an owning MPPI (device-resident rollouts)")
        return self._owner.get_cost()

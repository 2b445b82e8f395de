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
        """cost.py:13-22, on the device.  Called the way MPPI.get_cost does -- with the owner's own rollout tensors (or no
        arguments) -- it evaluates the device-resident rollouts in place; any other tensors (a subset, edited
        trajectories, another planner's rollouts) are uploaded and evaluated as given, like the reference does."""
        if self._owner is None:
            raise RuntimeError("Cost.evaluate_costs needs an owning MPPI (its device context evaluates the cost)")
        o = self._owner
        own = (all_traj is None and closest_dist_all is None) or \
              (all_traj is o.all_traj and closest_dist_all is o.closest_dist_all and o._generation > 0)
        if own:
            return o.get_cost()
        tr = torch.as_tensor(all_traj, dtype=torch.float32)
        if tr.ndim != 3 or tr.shape[1] != o.dt_H or tr.shape[2] != o.n_dof:
            raise ValueError(f"all_traj must be [B, {o.dt_H}, {o.n_dof}], got {tuple(tr.shape)}")
        o._push()
        return torch.from_numpy(o._engine.cost_eval(tr.numpy(), torch.as_tensor(closest_dist_all, dtype=torch.float32).numpy()))

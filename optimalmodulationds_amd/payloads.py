"""The two dictionaries the reference's processes exchange (pickled over ZMQ PUB/SUB, which stays out of scope): what the
planner publishes after an iteration (``ds_mppi/frankaPlanner.py:162-179``) and what the integrator publishes after a tick
(``ds_mppi/frankaIntegratorSwitching.py:111-117``).  Same keys, shapes and dtypes (torch CPU tensors), so that a process
built on this package can stand in for either side of the reference's deployment; the consumer of the policy dictionary is
``TensorPolicyMPPI.update_with_data`` (``policy.py:115-127``)."""
from __future__ import annotations

import torch

from .fk_num import numeric_fk_model, numeric_fk_model_vec


def kernel_fk(q, dh_params):
    """Visualisation points of one navigation kernel centre, appended to ``all_kernel_fk`` when the kernel is added
    (frankaPlanner.py:162-163): ``numeric_fk_model(q, dh_params, 2)[0][1:].flatten(0, 1)`` -> [(n-1)*2, 3]."""
    links, _ = numeric_fk_model(torch.as_tensor(q, dtype=torch.float32), dh_params, 2)
    return links[1:].flatten(0, 1)


def planner_payload(mppi, cost, all_kernel_fk=()):
    """frankaPlanner.py:166-177: the policy dictionary sent to the integrator after an iteration.  ``cost`` = the tensor
    ``mppi.get_cost()`` returned for the current rollouts; ``all_kernel_fk`` = the driver's list of ``kernel_fk`` entries."""
    K = mppi.Policy.n_kernels
    n = mppi.n_dof
    best_idx = int(torch.argmin(torch.as_tensor(cost)))
    best_traj_fk, _ = numeric_fk_model_vec(mppi.all_traj[best_idx:best_idx + 1].reshape(-1, n), mppi.dh_params, 2)
    return {'n_kernels': K,
            'mu_c': mppi.Policy.mu_c[0:K],
            'alpha_c': mppi.Policy.alpha_c[0:K],
            'sigma_c': mppi.Policy.sigma_c[0:K],
            'norm_basis': mppi.Policy.kernel_obstacle_bases[0:K],
            'kernel_fk': list(all_kernel_fk),
            'best_traj_fk': best_traj_fk.reshape(mppi.dt_H, -1, 3)[-1].unsqueeze(0)}


def integrator_state(mppi_step):
    """frankaIntegratorSwitching.py:114-116: the state dictionary sent to the planner after a tick."""
    return {'q': mppi_step.q_cur, 'dq': mppi_step.qdot[0, :], 'ds_idx': mppi_step.DS_idx}


def integrator_tick(mppi_step, policy_data, obstacles, dt_sim):
    """One tick of the fast loop (frankaIntegratorSwitching.py:102-113) on an N = 1, H = 2 ``MPPI``: install the planner's
    policy, take the obstacles, sample (alpha_s = 0 there), propagate, integrate and clamp ``q_cur``."""
    mppi_step.Policy.update_with_data(policy_data)
    if obstacles is not None:
        mppi_step.update_obstacles(obstacles)
    mppi_step.Policy.sample_policy()
    mppi_step.propagate()
    mppi_step.q_cur = mppi_step.q_cur + mppi_step.qdot[0, :] * dt_sim
    mppi_step.q_cur = torch.clamp(mppi_step.q_cur, mppi_step.Cost.q_min, mppi_step.Cost.q_max)
    return integrator_state(mppi_step)

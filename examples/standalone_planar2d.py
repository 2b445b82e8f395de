#!/usr/bin/env python3
"""The reference's 2-DoF stand-alone driver (ds_mppi/scripts/standalonePlanar2d.py:145-217) without
the matplotlib front end: an exploration MPPI (N rollouts x H steps) plus a 1-rollout stepping MPPI that
moves the robot with the planned policy, until the goal is reached; prints the reference's own
"Time per rollout step" figure."""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimalmodulationds_amd import MPPI, LinDS, RobotSdfCollisionNet, scenes  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(max_iter=400, n_traj=100, dt_h=10, quiet=False):
    DOF = 2
    nn_model = RobotSdfCollisionNet(in_channels=DOF + 3, out_channels=DOF, layers=[256] * 4, skips=[])
    nn_model.load_weights(os.path.join(ROOT, "tests", "golden", "weights", "planar2.npz"), {})
    q_0, q_f = torch.tensor([-3.14, 0.0]), torch.tensor([3.14, 0.0])
    dh = torch.tensor(scenes.planar_dh_params(DOF, 3.0))
    obs = torch.tensor(scenes.planar2_scene(2))
    DS = [LinDS(q_f), LinDS(q_0)]
    dt, dt_sim = 0.3, 0.1
    dst_thr, thr_rbf_add, thr_dot_add = 0.5, 0.05, -0.9
    mppi = MPPI(q_0, q_f, dh, obs, dt, dt_h, n_traj, DS, dh[:, 2], nn_model, 2)
    mppi.Policy.sigma_c_nominal = 0.5
    mppi.Policy.alpha_s = 2
    mppi.dst_thr = dst_thr / 2
    mppi.ker_thr = 1e-3
    mppi.ignored_links = []
    mppi.Cost.q_min = -0.99 * 3.14 * torch.ones(DOF)
    mppi.Cost.q_max = 0.99 * 3.14 * torch.ones(DOF)
    mppi_step = MPPI(q_0, q_f, dh, obs, dt_sim, 1, 1, DS, dh[:, 2], nn_model, 1)
    mppi_step.Policy.alpha_s *= 0
    mppi_step.ignored_links = []
    n_iter, t0 = 0, time.time()
    while torch.norm(mppi.q_cur - q_f) > 0.01 and n_iter < max_iter:
        mppi.Policy.sample_policy()
        all_traj, dist_all, kval, dots, _ = mppi.propagate()
        mppi.get_cost()
        mppi.shift_policy_means()
        cands = mppi.Policy.check_traj_for_kernels(all_traj, dist_all, dots, dst_thr - mppi.dst_thr, thr_rbf_add, thr_dot_add)
        if len(cands) > 0:
            norm, closest_idx = torch.norm(cands - mppi.q_cur, 2, -1).min(dim=0)
            idx_to_add = closest_idx if norm < 1e-1 else torch.randint(cands.shape[0], (1,))[0]
            t_i, h_i = mppi.Policy.last_candidate_index[idx_to_add]
            mppi.Policy.add_kernel(cands[idx_to_add], dist_all[t_i, h_i], mppi.norm_basis[int(t_i), int(h_i)])
        # move the robot with the planned policy (standalonePlanar2d.py:181-192)
        mppi_step.Policy.mu_c = mppi.Policy.mu_c
        mppi_step.Policy.sigma_c = mppi.Policy.sigma_c
        mppi_step.Policy.alpha_c = mppi.Policy.alpha_c
        mppi_step.Policy.n_kernels = mppi.Policy.n_kernels
        mppi_step.Policy.sample_policy()
        mppi_step.q_cur = copy.copy(mppi.q_cur)
        mppi_step.propagate()
        mppi.q_cur = mppi.q_cur + mppi_step.qdot[0, :] * dt_sim
        n_iter += 1
        if not quiet and n_iter % 20 == 0:
            print(f"Iteration:{n_iter:4d}, |q - qf| {float(torch.norm(mppi.q_cur - q_f)):.3f}, kernels {mppi.Policy.n_kernels}")
    td = time.time() - t0
    print('Time: ', td)
    print('Time per iteration: ', td / n_iter, 'Hz: ', 1 / (td / n_iter))
    print('Time per rollout step: ', td / (n_iter * n_traj * dt_h))
    return mppi, n_iter


if __name__ == "__main__":
    main()

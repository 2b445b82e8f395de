#!/usr/bin/env python3
"""The reference's planar 7-DoF stand-alone driver (ds_mppi/scripts/standalonePlanar7d.py:58-166) on the facade, without the
matplotlib front end: the dummy-sphere scene of :68-73 (three spheres + one far dummy), link length 1, an exploration MPPI
(20 rollouts x 10 steps, k = 1 closest obstacle, `:96-102`) and a 1-rollout stepping MPPI (`:104-108`) that moves the robot with
the planned policy (dt_sim = 0.02) until |q - q_f| < 0.1; kernels are added from the exploration rollouts as the script does
(`:130-147`); prints the reference's own "Time per rollout step" figure (`standalonePlanar2d.py:217`).  BASELINE configs[1] is
this robot at 1024 rollouts x 32 steps with 8 obstacles: `main(n_traj=1024, dt_h=32, n_extra_obs=4)`."""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimalmodulationds_amd import MPPI, LinDS, RobotSdfCollisionNet, scenes  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(max_iter=2000, n_traj=20, dt_h=10, n_extra_obs=0, quiet=False):
    DOF, L = 7, 1.0
    nn_model = RobotSdfCollisionNet(in_channels=DOF + 3, out_channels=DOF, layers=[256] * 4, skips=[])
    nn_model.load_weights(os.path.join(ROOT, "tests", "golden", "weights", "planar7.npz"), {})
    q_0, q_f = torch.zeros(DOF), torch.zeros(DOF)                       # :58-62
    q_0[0], q_f[0] = torch.pi / 2, -torch.pi / 2
    dh = torch.tensor(scenes.planar_dh_params(DOF, L))                  # :64-66
    obs = torch.tensor(scenes.planar7_scene(n_extra_obs))               # :68-73: three spheres + the far dummy (+ seeded extras)
    DS = [LinDS(q_f), LinDS(q_0)]                                       # :79-81
    dt, dt_sim = 0.3, 0.02                                              # :85-86
    dst_thr, thr_rbf_add, thr_dot_add = 0.5, 0.2, -0.9                  # :90-92
    mppi = MPPI(q_0, q_f, dh, obs, dt, dt_h, n_traj, DS, dh[:, 2], nn_model, 1)   # :95
    mppi.Policy.sigma_c_nominal = 0.5
    mppi.Policy.alpha_s = 0.75
    mppi.Policy.policy_upd_rate = 0.5                                   # an attribute nobody reads, kept like the script
    mppi.dst_thr = dst_thr / 2
    mppi.ker_thr = 1e-3
    mppi.ignored_links = []
    mppi_step = MPPI(q_0, q_f, dh, obs, dt_sim, 1, 1, DS, dh[:, 2], nn_model, 1)   # :104-108
    mppi_step.Policy.alpha_s *= 0
    mppi_step.ignored_links = []
    mppi_step.dst_thr = 1
    n_iter, t0 = 0, time.time()
    while torch.norm(mppi.q_cur - q_f) > 0.1 and n_iter < max_iter:     # :119
        mppi.Policy.sample_policy()
        all_traj, dist_all, kval, dots, _ = mppi.propagate()
        mppi.get_cost()
        mppi.shift_policy_means()
        cands = mppi.Policy.check_traj_for_kernels(all_traj, dist_all, dots, dst_thr - mppi.dst_thr, thr_rbf_add, thr_dot_add)   # :130-131
        if len(cands) > 0:                                              # :133-147
            norm, closest_idx = torch.norm(cands - mppi.q_cur, 2, -1).min(dim=0)
            idx_to_add = closest_idx if norm < 1e-1 else torch.randint(cands.shape[0], (1,))[0]
            t_i, h_i = mppi.Policy.last_candidate_index[idx_to_add]
            mppi.Policy.add_kernel(cands[idx_to_add], dist_all[t_i, h_i], mppi.norm_basis[int(t_i), int(h_i)])
        # move the robot with the planned policy (:150-161)
        mppi_step.Policy.mu_c = mppi.Policy.mu_c
        mppi_step.Policy.sigma_c = mppi.Policy.sigma_c
        mppi_step.Policy.alpha_c = mppi.Policy.alpha_c
        mppi_step.Policy.n_kernels = mppi.Policy.n_kernels
        mppi_step.Policy.sample_policy()
        mppi_step.q_cur = copy.copy(mppi.q_cur)
        mppi_step.propagate()
        mppi.q_cur = mppi.q_cur + mppi_step.qdot[0, :] * dt_sim
        n_iter += 1
        if not quiet and n_iter % 50 == 0:
            print(f"Iteration:{n_iter:4d}, |q - qf| {float(torch.norm(mppi.q_cur - q_f)):.3f}, kernels {mppi.Policy.n_kernels}")
    td = time.time() - t0
    print('Time: ', td)
    print('Time per iteration: ', td / max(n_iter, 1), 'Hz: ', 1 / (td / max(n_iter, 1)))
    print('Time per rollout step: ', td / (max(n_iter, 1) * n_traj * dt_h))
    return mppi, n_iter


if __name__ == "__main__":
    main()

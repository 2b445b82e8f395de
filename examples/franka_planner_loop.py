#!/usr/bin/env python3
"""Headless planner loop on the MI355X path -- the control flow of the reference's slow loop
(ds_mppi/frankaPlanner.py:99-189) with the ZMQ sockets replaced by in-process stand-ins: the
"integrator" advances q along the weighted rollout velocity, the "obstacle streamer" serves the
shelf scene (optionally translating it, obstacleStreamer.py:120-142).

    python examples/franka_planner_loop.py --iters 50 --rollouts 1024 --horizon 32
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimalmodulationds_amd import MPPI, LinDS, RobotSdfCollisionNet, scenes  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(iters=50, n_traj=1024, horizon=32, moving=False, weights=None, quiet=False):
    nn_model = RobotSdfCollisionNet(in_channels=10, out_channels=9, layers=[256] * 4, skips=[])
    nn_model.load_weights(weights or os.path.join(ROOT, "tests", "golden", "weights", "franka.npz"), {})
    nn_model.model_jit = nn_model
    nn_model.update_aot_lambda()
    q_0, q_f = torch.tensor(scenes.FRANKA_Q0), torch.tensor(scenes.FRANKA_QF)
    dh = torch.tensor(scenes.franka_dh_params())
    shelf = torch.tensor(scenes.shelf_scene())
    # config.yaml:40-58
    mppi = MPPI(q_0, q_f, dh, shelf, 0.5, horizon, n_traj, [LinDS(q_f), LinDS(q_0)], dh[:, 2], nn_model, 5)
    mppi.Policy.sigma_c_nominal = 1
    mppi.Policy.alpha_s = 3
    mppi.Policy.policy_upd_rate = 0.5
    mppi.Policy.p = 2
    mppi.dst_thr = 0.01
    mppi.ker_thr = 0.1
    dst_thr, thr_rbf_add, thr_dot_add = 0.03, 0.3, -0.9
    t0 = time.time()
    for it in range(iters):
        if moving:
            obs = shelf.clone()
            obs[:, 1] += 0.05 * np.sin(0.3 * it)
            mppi.update_obstacles(obs)
            mppi.update_kernel_normal_bases()
        mppi.Policy.sample_policy()
        all_traj, dist_all, kval, dots, _ = mppi.propagate()
        cost = mppi.get_cost()
        best_idx = torch.argmin(cost)
        _, n_upd = mppi.shift_policy_means()
        cands = mppi.Policy.check_traj_for_kernels(all_traj, dist_all, dots, dst_thr - mppi.dst_thr, thr_rbf_add, thr_dot_add)
        if len(cands) > 0:
            norm, closest_idx = torch.norm(cands - mppi.q_cur, 2, -1).min(dim=0)
            idx_to_add = closest_idx if norm < 1e-1 else torch.randint(cands.shape[0], (1,))[0]
            t_i, h_i = mppi.Policy.last_candidate_index[idx_to_add]
            mppi.Policy.add_kernel(cands[idx_to_add], dist_all[t_i, h_i], mppi.norm_basis[int(t_i), int(h_i)])
        # stand-in for the integrator process: follow the weighted rollout velocity
        mppi.q_cur = mppi.q_cur + mppi.get_qdot('weighted') * 0.05
        if not quiet:
            print(f"Iteration:{it + 1:4d}, best cost {float(cost[best_idx]):8.3f}, updated {n_upd:2d}, "
                  f"kernels {mppi.Policy.n_kernels:2d}, |q - qf| {float(torch.norm(mppi.q_cur - q_f)):.3f}")
    td = time.time() - t0
    print(f"Time per iteration: {td / iters * 1e3:.2f} ms, time per rollout step: {td / (iters * n_traj * horizon) * 1e9:.1f} ns "
          f"({iters * n_traj * horizon / td:,.0f} rollout-steps/s incl. host transfers of all rollout tensors)")
    return mppi


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--rollouts", type=int, default=1024)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--moving", action="store_true")
    ap.add_argument("--weights", default=None, help=".pt checkpoint of the reference or .npz export")
    a = ap.parse_args()
    main(a.iters, a.rollouts, a.horizon, a.moving, a.weights)

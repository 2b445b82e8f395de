#!/usr/bin/env python3
"""Headless planner loop on the MI355X path -- the control flow of the reference's slow loop
(ds_mppi/frankaPlanner.py:99-189) with the ZMQ sockets replaced by in-process stand-ins: the
"integrator" advances q along the weighted rollout velocity, the "obstacle streamer" serves the
configured scene (optionally translating it, obstacleStreamer.py:120-142).

    python examples/franka_planner_loop.py --iters 50 --rollouts 1024 --horizon 32
    python examples/franka_planner_loop.py --config examples/config.yaml          # the reference's YAML keys (its own config.yaml works too)

The loop body is the reference's, statement for statement, except for ONE line: the reference finds the rollout a kernel candidate
came from by comparing the candidate with every state of ``all_traj`` (frankaPlanner.py:159); here the candidate search (on the
device) reports that index itself (``Policy.last_candidate_index``), so ``all_traj`` never has to leave the GPU -- the rollout tensors
``propagate()`` returns are fetched row-wise when indexed (``MPPI(..., lazy_rollouts=True)``, optimalmodulationds_amd/lazy.py; the
class's default hands out plain torch tensors like the reference, one 0.4 ms fetch per propagate at 1024 x 32)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimalmodulationds_amd import MPPI, LinDS, RobotSdfCollisionNet, scenes  # noqa: E402
from optimalmodulationds_amd.fk_num import numeric_fk_model, numeric_fk_model_vec  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# config.yaml:40-58 + :1-5, :30-31 -- what the loop uses when no --config is given
DEFAULTS = {"collision_model": {"fname": "franka_collision_model.pt", "closest_spheres": 5, "obstacle": "shelf"},
            "general": {"q_0": list(map(float, scenes.FRANKA_Q0)), "q_f": list(map(float, scenes.FRANKA_QF))},
            "planner": {"n_trajectories": 40, "horizon": 10, "dt": 0.5, "collision_threshold": 0.01, "kernel_width": 1, "kernel_p": 2,
                        "alpha_sampling_sigma": 3, "policy_update_rate": 0.5, "kernel_update_threshold": 0.1, "update_kernel_bases": False,
                        "kernel_adding_collision_thr": 0.03, "kernel_adding_dotproduct_thr": -0.9, "kernel_adding_kernels_thr": 0.3}}


def load_config(path=None):
    """The reference's YAML (frankaPlanner.py:20-21); sections / keys it lacks fall back to the reference's defaults."""
    cfg = {k: dict(v) for k, v in DEFAULTS.items()}
    if path:
        import yaml
        with open(path) as f:
            user = yaml.safe_load(f) or {}
        for sec in cfg:
            cfg[sec].update(user.get(sec) or {})
    return cfg


def weights_file(fname, config_dir=None):
    """collision_model.fname: where the reference looks ('../mlp_learn/models/' + fname, frankaPlanner.py:45), then as given, then
    the repo's export of the same checkpoint (tests/golden/weights/franka.npz <- franka_collision_model.pt == franka_256x5.pt)."""
    cands = [fname]
    if config_dir:
        cands += [os.path.join(config_dir, "..", "mlp_learn", "models", fname), os.path.join(config_dir, fname)]
    for c in cands:
        if os.path.exists(c):
            return c
    if os.path.basename(fname) in ("franka_collision_model.pt", "franka_256x5.pt"):
        return os.path.join(ROOT, "tests", "golden", "weights", "franka.npz")
    raise FileNotFoundError(f"collision_model.fname = {fname!r} not found (tried {cands})")


def main(iters=50, n_traj=None, horizon=None, moving=False, weights=None, quiet=False, config=None):
    cfg = load_config(config)
    pl = cfg["planner"]
    DOF = 7
    nn_model = RobotSdfCollisionNet(in_channels=DOF + 3, out_channels=9, layers=[256] * 4, skips=[])
    nn_model.load_weights(weights or weights_file(cfg["collision_model"]["fname"], os.path.dirname(os.path.abspath(config)) if config else None), {})
    nn_model.model_jit = nn_model
    nn_model.update_aot_lambda()
    q_0, q_f = torch.tensor(cfg["general"]["q_0"], dtype=torch.float32), torch.tensor(cfg["general"]["q_f"], dtype=torch.float32)
    dh_params = torch.tensor(scenes.franka_dh_params())
    scene = torch.tensor(scenes.STREAMED_SCENES[cfg["collision_model"]["obstacle"]]())
    N_traj = int(n_traj or pl["n_trajectories"])
    dt_H = int(horizon or pl["horizon"])
    # frankaPlanner.py:72-88
    dst_thr = pl["kernel_adding_collision_thr"]
    thr_rbf_add = pl["kernel_adding_kernels_thr"]
    thr_dot_add = pl["kernel_adding_dotproduct_thr"]
    mppi = MPPI(q_0, q_f, dh_params, scene, pl["dt"], dt_H, N_traj, [LinDS(q_f), LinDS(q_0)], dh_params[:, 2], nn_model,
                int(cfg["collision_model"]["closest_spheres"]), lazy_rollouts=True)
    mppi.Policy.sigma_c_nominal = pl["kernel_width"]
    mppi.Policy.alpha_s = pl["alpha_sampling_sigma"]
    mppi.Policy.policy_upd_rate = pl["policy_update_rate"]
    mppi.Policy.p = pl["kernel_p"]
    mppi.dst_thr = pl["collision_threshold"]
    mppi.ker_thr = pl["kernel_update_threshold"]
    all_kernel_fk = []
    t0 = time.time()
    for it in range(iters):
        if moving:
            obs = scene.clone()
            obs[:, 1] += 0.05 * np.sin(0.3 * it)
            mppi.update_obstacles(obs)
        if pl["update_kernel_bases"] or moving:
            mppi.update_kernel_normal_bases()
        mppi.Policy.sample_policy()
        all_traj, closests_dist_all, kernel_val_all, dotproducts_all, _ = mppi.propagate()
        cost = mppi.get_cost()
        best_idx = torch.argmin(cost)
        _, n_upd = mppi.shift_policy_means()
        kernel_candidates = mppi.Policy.check_traj_for_kernels(all_traj, closests_dist_all, dotproducts_all, dst_thr - mppi.dst_thr, thr_rbf_add,
                                                               thr_dot_add)
        if len(kernel_candidates) > 0:
            rand_idx = torch.randint(kernel_candidates.shape[0], (1,))[0]
            closest_candidate_norm, closest_idx = torch.norm(kernel_candidates - mppi.q_cur, 2, -1).min(dim=0)
            idx_to_add = closest_idx if closest_candidate_norm < 1e-1 else rand_idx
            idx_i, idx_h = mppi.Policy.last_candidate_index[idx_to_add]      # (the reference: torch.where((all_traj == candidate).all(-1)))
            mppi.Policy.add_kernel(kernel_candidates[idx_to_add], closests_dist_all[idx_i, idx_h], mppi.norm_basis[idx_i, idx_h].squeeze())
            kernel_fk, _ = numeric_fk_model(kernel_candidates[idx_to_add], dh_params, 2)
            all_kernel_fk.append(kernel_fk[1:].flatten(0, 1))
        # the best trajectory's link points (the visualisation payload of frankaPlanner.py:166-168): one rollout's rows
        best_traj_fk, _ = numeric_fk_model_vec(mppi.all_traj[best_idx:best_idx + 1].view(-1, DOF), dh_params, 2)
        # stand-in for the integrator process: follow the weighted rollout velocity
        mppi.q_cur = mppi.q_cur + mppi.get_qdot('weighted') * 0.05
        if not quiet:
            print(f"Iteration:{it + 1:4d}, best cost {float(cost[best_idx]):8.3f}, updated {n_upd:2d}, "
                  f"kernels {mppi.Policy.n_kernels:2d}, |q - qf| {float(torch.norm(mppi.q_cur - q_f)):.3f}")
    td = time.time() - t0
    print(f"Time per iteration: {td / iters * 1e3:.2f} ms, time per rollout step: {td / (iters * N_traj * dt_H) * 1e9:.1f} ns "
          f"({iters * N_traj * dt_H / td:,.0f} rollout-steps/s through the reference-shaped classes, kernel count {mppi.Policy.n_kernels})")
    return mppi


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--rollouts", type=int, default=None, help="overrides planner.n_trajectories")
    ap.add_argument("--horizon", type=int, default=None, help="overrides planner.horizon")
    ap.add_argument("--moving", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--config", default=None, help="YAML with the reference's keys (ds_mppi/config.yaml); default: its values")
    ap.add_argument("--weights", default=None, help=".pt checkpoint of the reference or .npz export (overrides collision_model.fname)")
    a = ap.parse_args()
    main(a.iters, a.rollouts, a.horizon, a.moving, a.weights, a.quiet, a.config)

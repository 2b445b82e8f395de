#!/usr/bin/env python3
"""Golden vectors for MPPI.update_kernel_normal_bases (ds_mppi/functions/MPPI.py:284-304) by RUNNING THE REFERENCE
(container-only, like tools/make_golden.py, whose model loading and workarounds it reuses): K kernel centres on the
Franka shelf scene, obstacles moved, bases recomputed.  Stores inputs and outputs only -> tests/golden/bases_franka.npz."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (imports the reference)


def main():
    torch.manual_seed(21)
    rng = np.random.RandomState(21)
    nn_model = mg.load_model("franka")
    dh, dh_a = mg.robot_setup("franka")
    q0 = torch.from_numpy(np.asarray(mg.scenes.FRANKA_Q0, np.float32))
    qf = torch.from_numpy(np.asarray(mg.scenes.FRANKA_QF, np.float32))
    obs = torch.from_numpy(mg.scenes.shelf_scene())
    N, H, k, K = 8, 2, 5, 6
    with mg.quiet():
        mppi = mg.MPPI(q0, qf, dh, obs, 0.5, H, N, [mg.LinDS(qf), mg.LinDS(q0)], dh_a, nn_model, k)
    mppi.dst_thr = 0.01
    mg.set_policy_state(mppi, K, rng, q0.numpy(), qf.numpy(), 1.0)
    obs2 = obs.clone()
    obs2[:, 1] += 0.03
    obs2[:, 2] -= 0.02
    mppi.update_obstacles(obs2)
    with mg.quiet():
        mppi.update_kernel_normal_bases()
        dist, grad = mppi.distance_repulsion_nn(mppi.Policy.mu_c[0:K], aot=False)
    fx = {"K": K, "k": k, "obs": mg.t2n(obs2), "mu_c": mg.t2n(mppi.Policy.mu_c[:K]),
          "ignored_links": np.asarray(mppi.ignored_links, np.int32),
          "bases": mg.t2n(mppi.Policy.kernel_obstacle_bases[:K]), "distance": mg.t2n(dist), "nn_grad": mg.t2n(grad),
          "dh_params": mg.t2n(dh), "q0": mg.t2n(q0), "qf": mg.t2n(qf)}
    path = os.path.join(mg.OUT, "bases_franka.npz")
    np.savez_compressed(path, **fx)
    B = fx["bases"]
    print(path, os.path.getsize(path), "bytes; orthonormality", np.abs(np.einsum('kij,kil->kjl', B, B) - np.eye(7)).max())


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden vectors for the Jacobian entry points of the reference's RobotSdfCollisionNet (mlp_learn/sdf/robot_sdf.py:68-110,
139-158: compute_signed_distance_wgrad with 'all' / a column list / 'closest', compute_signed_distance_wgrad2, dist_grad_closest),
written by RUNNING THE REFERENCE.  Container-only (imports /root/reference); stores inputs and outputs only.

    python tools/make_golden_wgrad.py        ->  tests/golden/wgrad_<kind>.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (the generator's loaders; importing it imports the reference)

ORDER = {"franka": [8, 0, 3, 1, 4, 2, 7, 5, 6], "planar7": [6, 5, 4, 3, 2, 1, 0], "franka_tanh": [8, 0, 3, 1, 4, 2, 7, 5, 6]}
COLS = {"franka": [0, 3, 8], "planar7": [1, 6], "franka_tanh": [2, 7]}


def vectors(kind, seed=29):
    nn_model = mg.load_model(kind)
    _, dof, out = mg.MODELS[kind]
    torch.manual_seed(seed)
    B = 40
    x = torch.empty(B, dof + 3).uniform_(-2.5, 2.5)
    x[:, dof:] = torch.empty(B, 3).uniform_(-0.2, 1.0) if kind.startswith("franka") else torch.empty(B, 3).uniform_(-7, 7)
    fx = {"x": mg.t2n(x)}
    d, g, mi = nn_model.compute_signed_distance_wgrad(x.clone(), "all")
    fx["all_dist"], fx["all_grads"] = mg.t2n(d), mg.t2n(g)
    d, g, mi = nn_model.compute_signed_distance_wgrad(x.clone(), COLS[kind])
    fx["cols"], fx["cols_grads"] = np.asarray(COLS[kind], np.int32), mg.t2n(g)
    d, g, mi = nn_model.compute_signed_distance_wgrad(x.clone(), "closest")
    fx["closest_dist"], fx["closest_grads"], fx["closest_idx"] = mg.t2n(d), mg.t2n(g), mg.t2n(mi).astype(np.int32)
    if dof == 7 and out == 7:        # compute_signed_distance_wgrad2 hard-codes a 7 x B x 7 cotangent (robot_sdf.py:145)
        d, g, mi = nn_model.compute_signed_distance_wgrad2(x.clone()[:, :])
        fx["w2_dist"], fx["w2_grads"], fx["w2_idx"] = mg.t2n(d), mg.t2n(g), mg.t2n(mi).astype(np.int32)
    nn_model.allocate_gradients(B - 8, mg.PARAMS)            # dist_grad_closest truncates to maxInputSize (robot_sdf.py:118-119)
    d, g, mi = nn_model.dist_grad_closest(x.clone())
    fx["dgc_dist"], fx["dgc_grads"], fx["dgc_idx"] = mg.t2n(d), mg.t2n(g), mg.t2n(mi).astype(np.int32)
    # a permuted link order: columns, arg-min and Jacobian columns are those of the re-ordered outputs (robot_sdf.py:90-91)
    nn_model.set_link_order(ORDER[kind])
    fx["order"] = np.asarray(ORDER[kind], np.int32)
    d, g, mi = nn_model.compute_signed_distance_wgrad(x.clone(), "closest")
    fx["ord_closest_dist"], fx["ord_closest_grads"], fx["ord_closest_idx"] = mg.t2n(d), mg.t2n(g), mg.t2n(mi).astype(np.int32)
    d, g, mi = nn_model.compute_signed_distance_wgrad(x.clone(), COLS[kind])
    fx["ord_cols_grads"] = mg.t2n(g)
    # how close each row's hidden pre-activations come to zero (rows whose ReLU mask may flip under fp32 rounding)
    feats = torch.cat((x, torch.sin(x), torch.cos(x)), dim=-1)
    zmin = torch.full((B,), 1e9)
    h = feats
    seq = nn_model.model.layers[0]
    for li in range(len(seq) - 1):
        z = seq[li][0](h)
        zmin = torch.minimum(zmin, z.abs().min(dim=1)[0].detach())
        h = seq[li][1](z)
    fx["min_abs_preact"] = mg.t2n(zmin)
    path = os.path.join(mg.OUT, f"wgrad_{kind}.npz")
    np.savez_compressed(path, **fx)
    print(f"wgrad_{kind}: {os.path.getsize(path) / 1024:.1f} KB, |J| max {np.abs(fx['all_grads']).max():.3f}, min|z| {float(zmin.min()):.2e}")


if __name__ == "__main__":
    for k in ("franka", "planar7", "franka_tanh"):
        vectors(k)

#!/usr/bin/env python3
"""Golden vectors for SDF training by RUNNING THE REFERENCE's model class and optimizer (container-only, like
tools/make_golden.py).  mlp_learn/train_sdf.py is a script that trains on import (and needs datasets that are not shipped), so
its epoch loop (train_sdf.py:96-113: full-batch forward, F.mse_loss, backward, Adam(lr = 2e-4) step) is restated here around the
reference's own ``RobotSdfCollisionNet(...).model`` (MLPRegression with NeRF features) and ``torch.optim.Adam``; the AMP
autocast / GradScaler of the script are no-ops on the CPU.  The data set is synthetic, in the script's 2-D toy layout
(train_sdf.py:38-44: x = [q, point], y = distances): a planar 2-link arm (link length 3, standalonePlanar2d.py:67-69), the
distance from a point to each of its two links.  Stored: x, y, the initial weights, the loss of every epoch, the weights after
10 and after 100 epochs -> tests/golden/train_sdf_planar2.npz.
Usage:  MPLBACKEND=Agg python tools/make_golden_train.py"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
REF = "/root/reference/python_scripts"
sys.path[:0] = [REF + "/mlp_learn"]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.nn.functional as F

from sdf.robot_sdf import RobotSdfCollisionNet  # noqa: E402  (reference)

from optimalmodulationds_amd.trainer import planar_link_distances  # noqa: E402  (the synthetic data set: data, not reference code)


def main():
    torch.manual_seed(20)
    rng = np.random.RandomState(20)
    B, width, n_hidden, epochs = 2048, 128, 4, 100
    x, y = planar_link_distances(rng, B, n_links=2, link_len=3.0)
    net = RobotSdfCollisionNet(in_channels=x.shape[1], out_channels=y.shape[1], skips=[], layers=[width] * n_hidden)
    model = net.model
    model.train()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    W0 = [sd0[f"layers.0.{i}.0.weight"].numpy() for i in range(n_hidden + 1)]
    b0 = [sd0[f"layers.0.{i}.0.bias"].numpy() for i in range(n_hidden + 1)]
    optimizer = torch.optim.Adam(model.parameters(), lr=2e-4)                     # train_sdf.py:84
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    losses, snaps = [], {}
    for e in range(epochs):                                                       # train_sdf.py:96-113
        y_pred = model.forward(xt)
        train_loss = F.mse_loss(y_pred, yt, reduction='mean')
        train_loss.backward()
        optimizer.step()
        optimizer.zero_grad()
        losses.append(train_loss.item())
        if e + 1 in (10, epochs):
            sd = model.state_dict()
            snaps[e + 1] = ([sd[f"layers.0.{i}.0.weight"].detach().numpy().copy() for i in range(n_hidden + 1)],
                            [sd[f"layers.0.{i}.0.bias"].detach().numpy().copy() for i in range(n_hidden + 1)])
    model.eval()
    with torch.no_grad():
        val = F.mse_loss(model.forward(xt), yt, reduction='mean').item()
    out = dict(x=x, y=y, losses=np.asarray(losses, np.float64), final_eval=np.float64(val), lr=np.float64(2e-4),
               epochs=np.int64(epochs), width=np.int64(width))
    for i in range(n_hidden + 1):
        out[f"W0_{i}"], out[f"b0_{i}"] = W0[i], b0[i]
        out[f"W10_{i}"], out[f"b10_{i}"] = snaps[10][0][i], snaps[10][1][i]
        out[f"W100_{i}"], out[f"b100_{i}"] = snaps[epochs][0][i], snaps[epochs][1][i]
    path = os.path.join(REPO, "tests", "golden", "train_sdf_planar2.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {os.path.getsize(path) / 1e3:.0f} KB; loss {losses[0]:.5f} -> {losses[-1]:.5f}")


if __name__ == "__main__":
    main()

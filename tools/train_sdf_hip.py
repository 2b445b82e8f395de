#!/usr/bin/env python3
"""SDF training on the MI355X -- the loop of the reference's mlp_learn/train_sdf.py:96-151 with its per-epoch work (full-batch
forward, MSE, backward, Adam) on the device (optimalmodulationds_amd.trainer.SdfTrainer -> csrc/train.hip) and its O(1) parts
(train / validation / test split, ReduceLROnPlateau, "save when the validation loss improves" with the same guards, the
checkpoint dictionary) on the host.

The reference trains on data sets that are not shipped (datasets/2d_toy_data.pt, %d_dof_data.pt: columns [q, point | link
distances]); without --data this script makes a synthetic one of the same layout (planar chain, point-to-link distances).

    python tools/train_sdf_hip.py --epochs 300 --rows 65536 --out gpurun_out/2dof_sdf_256x5_toy.pt
    python tools/train_sdf_hip.py --data my_data.npy --q-dof 7 --epochs 1000        # [B, q_dof + 3 + C] rows"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None, help=".npy [B, q_dof + point_dims + C]; default: synthetic planar data")
    ap.add_argument("--q-dof", type=int, default=2)
    ap.add_argument("--point-dims", type=int, default=2)
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--hidden", type=int, default=4, help="hidden layers (train_sdf.py: s = 256, n_layers - 1 = 4)")
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--init", default=None, help="checkpoint (.pt / .npz) to continue from, like train_sdf.py:70")
    ap.add_argument("--out", default=None)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    import torch
    from optimalmodulationds_amd.trainer import ReduceLROnPlateau, SdfTrainer, checkpoint_dict, planar_link_distances
    rng = np.random.RandomState(args.seed)
    if args.data:
        data = np.load(args.data).astype(np.float32)
        nin = args.q_dof + args.point_dims
        x_all, y_all = data[:, :nin], data[:, nin:]
    else:
        x_all, y_all = planar_link_distances(rng, args.rows, n_links=args.q_dof, link_len=3.0)
    n = x_all.shape[0]
    n_train, n_val = int(n * 0.98), int(n * 0.001)                              # train_sdf.py:47-53
    x_tr, y_tr = x_all[:n_train], y_all[:n_train]
    x_va, y_va = x_all[n_train:n_train + max(n_val, 1)], y_all[n_train:n_train + max(n_val, 1)]
    d, C = x_all.shape[1], y_all.shape[1]
    dims = [3 * d] + [args.width] * args.hidden + [C]
    torch.manual_seed(args.seed)
    if args.init:
        from optimalmodulationds_amd import RobotSdfCollisionNet
        nn = RobotSdfCollisionNet(d, C, [], [args.width] * args.hidden)
        nn.load_weights(args.init, {'device': 'cpu', 'dtype': torch.float32})
        W, b = nn.model.W, nn.model.b
    else:                                                                        # torch's default nn.Linear initialisation
        lin = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
        W, b = [l.weight.detach().numpy() for l in lin], [l.bias.detach().numpy() for l in lin]
    train = SdfTrainer(dims, "relu")          # ONE trainer holds both splits: the validation pass needs no weight copy
    train.set_weights(W, b)
    train.set_data(x_tr, y_tr)
    train.set_val_data(x_va, y_va)
    sched = ReduceLROnPlateau(args.lr, factor=0.5, patience=5000, threshold=0.01, eps=1e-4)   # train_sdf.py:85-87
    min_loss, e_notsaved, t_dev = None, 0, 0.0
    close = y_va[:, -1] < 1
    for e in range(args.epochs):
        t0 = time.time()
        train_loss = train.step(lr=sched.lr)
        t_dev += time.time() - t0
        val_loss, pred = train.eval(want_pred=True, val=True)
        l1_close = float(np.abs(pred[close, -1] - y_va[close, -1]).mean()) if close.any() else float("nan")
        if e == 0:
            min_loss = val_loss
        sched.step(val_loss)
        e_notsaved += 1
        if val_loss < min_loss and e > 100 and e_notsaved > 100:                 # train_sdf.py:127
            e_notsaved, min_loss = 0, val_loss
            if args.out:
                Wc, bc = train.get_weights()
                torch.save(checkpoint_dict(e, Wc, bc, d, C, train.optimizer_state_dict()), args.out)   # train_sdf.py:130-138
                print("saving model", val_loss)
        if e % 20 == 0 or e == args.epochs - 1:
            print("Epoch: %d (Saved at %d), Train Loss: %4.3f, Validation Loss: %4.3f (%4.3f), Epoch time: %4.4f s, LR = %4.8f" % (
                e, e - e_notsaved + 1, train_loss, val_loss, l1_close, time.time() - t0, sched.lr))
    flops = 6.0 * n_train * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))   # forward + two backward GEMMs per layer
    print(f"{args.epochs} epochs on {n_train} rows: {1e3 * t_dev / args.epochs:.2f} ms per epoch on the device "
          f"({flops * args.epochs / t_dev / 1e12:.1f} TFLOP/s of the fp32 GEMMs)")


if __name__ == "__main__":
    main()

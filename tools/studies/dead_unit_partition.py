"""CPU study for the exact zero-skip of k_pass1: a hidden unit whose activation is exactly zero for every row of a tile
adds fmaf(0, w, acc) = acc to every chain of the next layer -- skipping it is EXACT whatever the summation order, so the tile may be
stored [A | B] (A = units that may fire, ascending; B = units presumed dead, ascending) and the next product stop after A, as long
as no unit of B fires in the tile (a "surprise": the tile is then re-stored in natural order and multiplied in full).
B is chosen at omds_set_mlp time from a SYNTHETIC sample (no scene known yet): units that fire for no sampled input.
This script: |A| per level for that choice, and on bench-like data (rollout states near the q0 -> qf path x the shelf scene, 64
consecutive pairs per tile) the share of tiles with a surprise per level and the k-chunks (of 8) multiplied.
    python tools/studies/dead_unit_partition.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import omds_oracle as orc                 # noqa: E402
from optimalmodulationds_amd import scenes            # noqa: E402


def levels(m, x):
    h = orc.positional_encoding(x)
    out = []
    for i in range(len(m.W) - 1):
        h = np.maximum(orc.linear(h, m.W[i], m.b[i]), 0)
        out.append(h > 0)
    return out


def synthetic_sample(n, n_dof, seed=12345, rows=4096):
    rng = np.random.RandomState(seed)
    q = rng.uniform(-np.pi, np.pi, (rows, n_dof))
    p = np.where(rng.rand(rows, 1) < 0.5, rng.uniform(-1.5, 1.5, (rows, 3)), rng.uniform(-8.0, 8.0, (rows, 3)))
    return np.concatenate([q, p], 1).astype(np.float32)


for kind, scene, q0, qf in (("franka", scenes.shelf_scene(), scenes.FRANKA_Q0, scenes.FRANKA_QF),):
    m = orc.Mlp.from_npz(os.path.join(ROOT, "tests", "golden", "weights", kind + ".npz"))
    n = m.W[0].shape[1] // 3 - 3
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    fired = [a.any(0) for a in levels(m, synthetic_sample(kind, n, rows=rows))]
    nA = [int(f.sum()) for f in fired]
    print(kind, "units that fire on the synthetic sample per level:", nA, "-> k-chunks of 8:", [(a + 7) // 8 for a in nA])
    rng = np.random.RandomState(0)
    T = 96
    s = rng.rand(T, 1).astype(np.float32)
    Q = (np.asarray(q0, np.float32) + s * (np.asarray(qf, np.float32) - np.asarray(q0, np.float32)) + 0.3 * rng.standard_normal((T, n))).astype(np.float32)
    O = scene.shape[0]
    x = np.concatenate([np.repeat(Q, O, 0), np.tile(scene[:, :3], (T, 1))], 1).astype(np.float32)   # rollout-major pairs
    acts = levels(m, x)
    nt = x.shape[0] // 64
    for L, a in enumerate(acts):
        tile_alive = a[:nt * 64].reshape(nt, 64, -1).any(1)            # [tiles, 256]
        surprise = tile_alive[:, ~fired[L]].any(1)
        print(f"  level {L}: |A| = {nA[L]}, tiles with a surprise {100 * surprise.mean():.3f} % ({int(surprise.sum())} of {nt}); units alive per tile: "
              f"mean {tile_alive.sum(1).mean():.0f}, max {tile_alive.sum(1).max()}")
    ch = [32 if L < 0 else (nA[L] + 7) // 8 for L in range(3)]
    print(f"  hidden-layer k-chunks multiplied: {sum(ch)} of 96; last layer chunks of 16: {(nA[3] + 15) // 16} of 16")

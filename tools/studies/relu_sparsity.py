"""How sparse the shipped distance network is under ReLU, and what a unit order buys the screening kernel (EXPERIMENTS.md C 4.1d): per
hidden layer the density of the activations over states x obstacles of the shelf task, the units that never fire, and the fraction
of 16-unit k-chunks that are zero for a whole block of 32 consecutive pairs -- natural order, sorted by firing frequency over all
pairs, sorted by the 8 nearest obstacles only.  numpy on the oracle (test infrastructure):  python tools/studies/relu_sparsity.py"""
import os
import numpy as np, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from oracle import omds_oracle as orc
from optimalmodulationds_amd import scenes
m = orc.Mlp.from_npz(os.path.join(ROOT, 'tests', 'golden', 'weights', 'franka.npz'))
obs = scenes.shelf_scene()
rng = np.random.RandomState(0)
q0, qf = np.array(scenes.FRANKA_Q0, np.float32), np.array(scenes.FRANKA_QF, np.float32)
T = 64
s = rng.rand(T,1).astype(np.float32)
Q = q0 + s*(qf-q0) + 0.3*rng.standard_normal((T,7)).astype(np.float32)
acts = [[] for _ in range(4)]
for t in range(T):
    x = np.concatenate([np.repeat(Q[t:t+1], obs.shape[0], 0), obs[:, :3]], 1).astype(np.float32)
    h = orc.positional_encoding(x)
    for i in range(4):
        h = np.maximum(h @ m.W[i].T + m.b[i], 0)
        acts[i].append(h > 0)
for i in range(4):
    A = np.stack(acts[i])            # [T, O, 256]
    dens = A.mean()
    O = A.shape[1]
    nb = O // 32
    blk = A[:, :nb*32].reshape(T, nb, 32, 256).any(2)        # [T, nb, 256] unit alive in block
    alive_frac = blk.mean()
    # static permutation: sort units by how often they are alive in a block
    order = np.argsort(blk.mean((0,1)))
    b2 = blk[:, :, order].reshape(T, nb, 16, 16).any(3)      # 16 slices of 16 units
    print(f"layer {i+1}: density {dens:.3f}, unit alive in a 32-pair block {alive_frac:.3f}, 16-unit slices with any alive unit (best static order) {b2.mean():.3f}; natural order {blk.reshape(T,nb,16,16).any(3).mean():.3f}")
    # per-rollout (all obstacles) dead units
    ro = A.any(1)   # [T,256]
    print(f"          units alive for at least one obstacle of the rollout: {ro.mean():.3f}")
print("---- per-position alive rate with static order (16-unit chunks), and 64-unit groups")
for i in range(4):
    A = np.stack(acts[i]); O = A.shape[1]; nb = O // 32
    blk = A[:, :nb*32].reshape(T, nb, 32, 256).any(2)
    order = np.argsort(-blk.mean((0,1)))   # most alive first
    b = blk[:, :, order]
    c16 = b.reshape(T, nb, 16, 16).any(3).mean((0,1))
    g64 = b.reshape(T, nb, 4, 64).any(3).mean((0,1))
    print(f"layer {i+1}: chunk alive rates {np.round(c16,2)}  mean {c16.mean():.3f}; group-of-64 alive {np.round(g64,2)} mean {g64.mean():.3f}")
    # tile of 256 pairs (8 blocks): all 8 waves dead
    nt = O // 256
    tb = A[:, :nt*256].reshape(T, nt, 256, 256).any(2)[:, :, order]
    print(f"          256-pair tiles: chunk alive mean {tb.reshape(T,nt,16,16).any(3).mean():.3f}")
print("---- ordering from the 8 nearest obstacles of each state vs from all pairs: chunks (>= 8) dead for a 32-pair block")
# nearest: by true min distance
D = []
for t in range(T):
    x = np.concatenate([np.repeat(Q[t:t+1], obs.shape[0], 0), obs[:, :3]], 1).astype(np.float32)
    y = orc.mlp_forward(m, x)/100.0 - obs[:,3:4]
    y[:, :3] = 1e6
    D.append(y.min(1))
D = np.stack(D)
near = np.argsort(D, 1)[:, :8]
for i in range(4):
    A = np.stack(acts[i]); O = A.shape[1]; nb = O//32
    cnt_all = A.reshape(-1,256).sum(0)
    cnt_near = np.stack([A[t, near[t]] for t in range(T)]).reshape(-1,256).sum(0)
    blk = A[:, :nb*32].reshape(T, nb, 32, 256).any(2)
    for name, cnt in (("all pairs", cnt_all), ("8 nearest", cnt_near)):
        order = np.argsort(-cnt, kind="stable")
        c16 = blk[:, :, order].reshape(T, nb, 16, 16).any(3)
        dead = 1 - c16[:, :, 8:].mean()*0.5 - 0.5   # fraction of all 16 chunks that are dead & tested
        print(f"layer {i+1} {name:10s}: dead chunk fraction {1-c16.mean():.3f} (of which in chunks >= 8: {(1-c16[:,:,8:]).sum()/c16.size:.3f}); never-fired units {int((cnt==0).sum())}")

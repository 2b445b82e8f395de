import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from helpers import weights_path
from oracle import omds_oracle as orc
from optimalmodulationds_amd import _lib as L, scenes
from test_gpu_sparse import _engine
m = orc.Mlp.from_npz(weights_path("planar7"))
rng = np.random.RandomState(3)
N=700; n=7
obs = np.concatenate([scenes.planar7_scene(4), np.c_[rng.uniform(-7, 7, (120, 2)), np.zeros(120), np.full(120, 0.5)]]).astype(np.float32)
near = (0.8 * rng.standard_normal((N, n))).astype(np.float32)
sp, de = _engine(m, N, obs), _engine(m, N, obs, flags=L.FLAG_DENSE_PASS1)
print(sp.pass1_skip_stats())
a=sp.dist_grad(near, want_mindist=True, want_idx=True)[2]; b=de.dist_grad(near, want_mindist=True, want_idx=True)[2]
_,_,mo,_=orc.distance_repulsion_nn(m, near, obs, 5, [])
print('sparse==dense', np.mean(a==b), 'sparse==oracle', np.mean(a==mo), 'dense==oracle', np.mean(b==mo))
bad=np.argwhere(a!=b); print(len(bad), bad[:10], 'rows (pair idx):', (bad[:10,0]*128+bad[:10,1]))
pi=bad[:,0]*128+bad[:,1]
print('tile64 ids', np.unique(pi//64)[:20], 'n tiles', len(np.unique(pi//64)), 'max diff', np.abs(a-b).max())
print(sp.pass1_skip_stats())
for N2 in (64, 256, 511, 512):
    s2, d2 = _engine(m, N2, obs), _engine(m, N2, obs, flags=L.FLAG_DENSE_PASS1)
    x=s2.dist_grad(near[:N2], want_mindist=True)[2]; y=d2.dist_grad(near[:N2], want_mindist=True)[2]
    print(N2, N2*128, 'equal', np.mean(x==y)); s2.close(); d2.close()
far = rng.uniform(-np.pi, np.pi, (N, n)).astype(np.float32)
wild_obs = np.c_[rng.uniform(-9, 9, (obs.shape[0], 3)), np.full(obs.shape[0], 0.1)].astype(np.float32)
for name,(q,scene) in dict(far=(far,obs), wild=(far,wild_obs)).items():
    sp.set_obstacles(scene); de.set_obstacles(scene)
    s0=sp.pass1_skip_stats()['surprises']
    a=sp.dist_grad(q, want_mindist=True)[2]; b=de.dist_grad(q, want_mindist=True)[2]
    _,_,mo,_=orc.distance_repulsion_nn(m, q[:128], scene, 5, [])
    bad=np.argwhere(a!=b); pi=bad[:,0]*128+bad[:,1]
    print(name,'sparse==dense', np.mean(a==b), 'sparse==oracle', np.mean(a[:128]==mo), 'dense==oracle', np.mean(b[:128]==mo), 'surprises', sp.pass1_skip_stats()['surprises']-s0,
          'bad tiles', len(np.unique(pi//64)), 'of', N*128//64, 'maxdiff', np.abs(a-b).max() if len(bad) else 0, 'first bad pairs', pi[:8], 'bad rows within tile', np.unique(pi%64)[:20])

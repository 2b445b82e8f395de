import sys, time, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from optimalmodulationds_amd import scenes, _lib
from optimalmodulationds_amd.engine import Engine
from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
from oracle import omds_oracle as orc
m = orc.Mlp.from_npz(os.path.join(ROOT, 'tests', 'golden', 'weights', 'franka.npz'))
obs = scenes.shelf_scene()
def run(N, H, flags, K=6, k=5):
    e = Engine(7, N, H, k, max_obs=512, flags=flags)
    e.set_mlp(m.W, m.b); e.set_obstacles(obs); e.set_screening(0)
    e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.5, 0.01, 0b111
    e.push_params(); e.set_ds(scenes.FRANKA_QF)
    e.set_cost(scenes.franka_dh_params(), np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32))
    rng = np.random.RandomState(0)
    mu_c = (scenes.FRANKA_Q0 + 0.2*rng.standard_normal((K,7))).astype(np.float32); sg_c = np.ones(K, np.float32); al_c = rng.standard_normal((K,7)).astype(np.float32)
    e.sample_policy(mu_c, sg_c, al_c, 0, 0, 3.0, K, seed=3)
    e.propagate(scenes.FRANKA_Q0)
    r = e.get_rollouts()
    e.propagate(scenes.FRANKA_Q0); e.sync()
    t0 = time.perf_counter()
    for _ in range(20): e.propagate(scenes.FRANKA_Q0)
    e.sync(); dt = (time.perf_counter()-t0)/20
    e.close()
    return r, dt
import sys as _s
for N, H in ([(int(a), 10) for a in _s.argv[1:]] or ((1024, 32), (128, 32), (40, 8), (4096, 8), (7, 3))):
    a, ta = run(N, H, 0)
    b, tb = run(N, H, _lib.FLAG_TAIL_FORWARD)
    same = all(np.array_equal(a[k], b[k]) for k in a)
    print(N, H, 'bit-identical' if same else 'DIFFERENT', f'emit {ta*1e3:.3f} ms  tail-forward {tb*1e3:.3f} ms per propagate', {k: float(np.abs(a[k]-b[k]).max()) for k in a if not np.array_equal(a[k], b[k])})

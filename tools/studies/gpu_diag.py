"""Diagnostic (GPU box): teacher-forced per-step errors of the HIP path vs golden, per scenario,
with the worst rollout's context printed."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from helpers import SCENARIOS, load, weights_path
from oracle import omds_oracle as orc
from test_gpu_parity import _engine

for name in SCENARIOS:
    fx = load(name)
    eng, m = _engine(fx, H=1)
    H = int(fx["H"]); dt = np.float32(fx["dt"]); k = int(fx["k"])
    worst = 0
    for it in range(int(fx["n_iter"])):
        pre = f"it{it}_"
        ref = fx[pre + "all_traj"]
        eng.set_policy_samples(fx[pre + "mu_tmp"], fx[pre + "sigma_tmp"], fx[pre + "alpha_tmp"])
        for i in range(1, H):
            q = np.ascontiguousarray(ref[:, i - 1, :])
            eng.propagate(q)
            r = eng.get_rollouts()
            nxt = q + dt * r["qdot"]
            e = np.abs(nxt - ref[:, i, :]).max(axis=1)
            t = int(e.argmax())
            if e[t] > 1e-5 * max(1, np.abs(ref).max()):
                o = orc.propagate(m, q, fx["qf"], fx["obs"], N=q.shape[0], H=1, dt=float(dt), k=k,
                                  ignored_links=fx["ignored_links"], mu_tmp=fx[pre + "mu_tmp"], sigma_tmp=fx[pre + "sigma_tmp"],
                                  alpha_tmp=fx[pre + "alpha_tmp"], prm=orc.Params(dst_thr=float(fx["dst_thr"])))
                d, g, mind, idx = eng.dist_grad(q, want_mindist=True, want_idx=True)
                od, og, omind, oidx = orc.distance_repulsion_nn(m, q, fx["obs"], k, fx["ignored_links"])
                sd = np.sort(omind[t])
                print(f"{name} it{it} step{i} rollout {t}: err {e[t]:.3e}  oracle-err {np.abs(q[t]+dt*o.qdot[t]-ref[t,i]).max():.2e}")
                print("   dist gpu/orc/ref", r["closest_dist_all"][t, 0], o.closest_dist_all[t, 0], fx[pre + "closest_dist_all"][t, i - 1])
                print("   dot  gpu/orc/ref", r["dot_products"][t, 0], o.dot_products[t, 0], fx[pre + "dot_products"][t, i - 1])
                print("   act  gpu/orc/ref", r["kernel_activations"][t, 0], o.kernel_activations[t, 0], fx[pre + "kernel_activations"][t, i - 1])
                print("   idx gpu", idx[t], "orc", oidx[t], " sorted d[k-1],d[k]:", sd[k - 1], sd[min(k, len(sd) - 1)])
                print("   grad gpu", g[t], "\n   grad orc", og[t])
                print("   |qdot| gpu", np.linalg.norm(r["qdot"][t]), "orc", np.linalg.norm(o.qdot[t]), "qdot diff", np.abs(r["qdot"][t] - o.qdot[t]).max())
            worst = max(worst, float(e.max()))
    print(f"{name}: worst next-state abs err {worst:.3e}")
    eng.close()

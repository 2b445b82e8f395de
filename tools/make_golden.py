#!/usr/bin/env python3
"""Generate the golden parity vectors under tests/golden/ by RUNNING THE REFERENCE.

Container-only tool: it imports the reference's Python modules from /root/reference (which
never travels to the GPU box), runs fixed-seed scenarios of the MPPI rollout + DS-modulation
path, and stores inputs (including the *sampled* policy tensors, because torch's CPU RNG
stream cannot be reproduced elsewhere) and all outputs as small .npz fixtures, plus the
network weights as flat fp32 arrays.  Nothing of the reference's source text is stored.

Two workarounds are applied, both documented in SURVEY.md section 0.4:
  * ``nn_model.aot_lambda = nn_model.functorch_vjp`` -- the reference's own eager fallback
    (robot_sdf.py:162) because aot_function(ts_compile) asserts under torch 2.10;
  * for the 2-DoF robot ``Cost.rest`` / ``q_min`` / ``q_max`` are set to 2-vectors like
    scripts/standalonePlanar2d.py:128-129 does (``rest`` is unused in the total cost).

Usage:  MPLBACKEND=Agg python tools/make_golden.py
"""
import contextlib
import io
import math
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
REF = "/root/reference/python_scripts"
sys.path[:0] = [REF + "/ds_mppi/functions", REF + "/mlp_learn"]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch

from MPPI import MPPI  # noqa: E402  (reference)
from LinDS import LinDS  # noqa: E402  (reference)
from sdf.robot_sdf import RobotSdfCollisionNet  # noqa: E402  (reference)

from optimalmodulationds_amd import scenes  # noqa: E402  (this repo: scene geometry only)

OUT = os.path.join(REPO, "tests", "golden")
PARAMS = {"device": "cpu", "dtype": torch.float32}
MODELS = {
    "franka": ("franka_collision_model.pt", 7, 9),
    "planar7": ("7dof_sdf_256x5_mesh.pt", 7, 7),
    "planar2": ("2dof_sdf_256x5_mesh.pt", 2, 2),
    # BASELINE.json config 3 names a "256-256-256 tanh" SDF; no such weights ship with the reference
    # (SURVEY 0.1), so this one is the reference's own MLPRegression class with act_fn=Tanh and seeded
    # synthetic weights -- it pins the tanh forward/backward arithmetic, not a trained model
    "franka_tanh": (None, 7, 9),
    # the reference's narrower shipped net (30-128-128-7): exercises the zero-padding of hidden layers to width 256
    "planar7_128": ("7dof_sdf_128x3_mesh.pt", 7, 7),
    # skip-connection layout (MLPRegression(..., skips=[2]), network_macros_mod.py:113-146): no such weights ship either;
    # the reference's own class with seeded synthetic weights pins the concatenation arithmetic, forward and backward
    "franka_skip": (None, 7, 9),
}


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def load_model(kind):
    fname, dof, out = MODELS[kind]
    if kind == "franka_skip":
        torch.manual_seed(20241)
        nn_model = RobotSdfCollisionNet(in_channels=dof + 3, out_channels=out, layers=[256] * 4, skips=[2])
        with torch.no_grad():
            last = nn_model.model.layers[-1][-1][0]
            last.weight.mul_(40.0)          # outputs are read as centimetres (C == 9): spread them
            last.bias.add_(12.0)
        nn_model.model.eval()
        nn_model.model.to(**PARAMS)
        try:
            nn_model.model_jit = torch.jit.optimize_for_inference(torch.jit.script(nn_model.model))
        except Exception as e:   # the drivers script the model (frankaPlanner.py:47); eager is the same arithmetic
            print("franka_skip: torch.jit.script failed (%s); eager model used" % type(e).__name__)
            nn_model.model_jit = nn_model.model
        nn_model.aot_lambda = nn_model.functorch_vjp
        return nn_model
    if fname is None:
        from sdf.network_macros_mod import MLPRegression
        from torch.nn import Tanh
        nn_model = RobotSdfCollisionNet(in_channels=dof + 3, out_channels=out, layers=[256] * 3, skips=[])
        torch.manual_seed(20240)
        nn_model.model = MLPRegression(dof + 3, out, [256] * 3, [], act_fn=Tanh, nerf=True)
        with torch.no_grad():
            last = nn_model.model.layers[0][-1][0]
            last.weight.mul_(40.0)          # outputs are read as centimetres (C == 9): spread them
            last.bias.add_(12.0)
        nn_model.model.eval()
        nn_model.model.to(**PARAMS)
        nn_model.model_jit = torch.jit.optimize_for_inference(torch.jit.script(nn_model.model))
        nn_model.aot_lambda = nn_model.functorch_vjp
        return nn_model
    layers = [128] * 2 if kind.endswith("_128") else [256] * 4
    nn_model = RobotSdfCollisionNet(in_channels=dof + 3, out_channels=out, layers=layers, skips=[])
    with quiet():
        nn_model.load_weights(REF + "/mlp_learn/models/" + fname, PARAMS)
    nn_model.model.to(**PARAMS)
    nn_model.model_jit = torch.jit.optimize_for_inference(torch.jit.script(nn_model.model))
    nn_model.aot_lambda = nn_model.functorch_vjp  # workaround 1
    return nn_model


def export_weights(kind, nn_model):
    sd = nn_model.model.state_dict()
    arrs = {}
    i, g, skip_after = 0, 0, []
    while f"layers.{g}.0.0.weight" in sd:      # modules in order; the encoded input is concatenated between modules
        j = 0
        while f"layers.{g}.{j}.0.weight" in sd:
            arrs[f"W{i}"] = sd[f"layers.{g}.{j}.0.weight"].numpy().astype(np.float32)
            arrs[f"b{i}"] = sd[f"layers.{g}.{j}.0.bias"].numpy().astype(np.float32)
            i += 1
            j += 1
        g += 1
        if f"layers.{g}.0.0.weight" in sd:
            skip_after.append(i - 1)
    if skip_after:
        arrs["skip_after"] = np.asarray(skip_after, np.int32)
    arrs["act"] = np.array("tanh" if kind.endswith("tanh") else "relu")
    os.makedirs(os.path.join(OUT, "weights"), exist_ok=True)
    np.savez(os.path.join(OUT, "weights", kind + ".npz"), **arrs)


def t2n(x):
    return x.detach().cpu().numpy().copy()


def robot_setup(kind):
    if kind.startswith("franka"):
        dh = torch.from_numpy(scenes.franka_dh_params())
        dh_a = dh[:, 2].clone()
    elif kind.startswith("planar7"):
        dh = torch.from_numpy(scenes.planar_dh_params(7, 1.0))
        dh_a = dh[:, 2].clone()
    else:
        dh = torch.from_numpy(scenes.planar_dh_params(2, 3.0))
        dh_a = dh[:, 2].clone()
    return dh, dh_a


def set_policy_state(mppi, K, rng, q0, qf, sigma_nom, alpha_scale=1.0):
    """Deterministic policy means: K kernel centres spread near the q0->qf segment."""
    P = mppi.Policy
    P.reset_policy()
    n = mppi.n_dof
    for kk in range(K):
        s = (kk + 0.5) / max(K, 1)
        c = q0 + s * (qf - q0) + 0.15 * rng.standard_normal(n).astype(np.float32)
        P.mu_c[kk] = torch.from_numpy(c.astype(np.float32))
        P.sigma_c[kk] = sigma_nom
        P.alpha_c[kk] = torch.from_numpy((alpha_scale * rng.standard_normal(n)).astype(np.float32))
    P.n_kernels = K


def run_scenario(name, kind, nn_model, *, N, H, dt, obs, k, q0, qf, dst_thr, ker_thr, alpha_s,
                 sigma_nom, K, n_iter=2, ignored_links=None, seed=0, q_cur=None, planar2_limits=False,
                 p=2, advance="best", ds_array=None, extra=None):
    torch.manual_seed(seed)
    rng = np.random.RandomState(seed + 1000)
    dh, dh_a = robot_setup(kind)
    q0_t = torch.from_numpy(np.asarray(q0, dtype=np.float32))
    qf_t = torch.from_numpy(np.asarray(qf, dtype=np.float32))
    obs_t = torch.from_numpy(np.asarray(obs, dtype=np.float32))
    DS_ARRAY = ds_array if ds_array is not None else [LinDS(qf_t), LinDS(q0_t)]
    with quiet():
        mppi = MPPI(q0_t, qf_t, dh, obs_t, dt, H, N, DS_ARRAY, dh_a, nn_model, k)
    mppi.Policy.sigma_c_nominal = sigma_nom
    mppi.Policy.alpha_s = alpha_s
    mppi.Policy.p = p
    mppi.dst_thr = dst_thr
    mppi.ker_thr = ker_thr
    if ignored_links is not None:
        mppi.ignored_links = list(ignored_links)
    if planar2_limits or kind == "planar2":  # workaround 2
        mppi.Cost.q_min = -0.99 * 3.14 * torch.ones(mppi.n_dof)
        mppi.Cost.q_max = 0.99 * 3.14 * torch.ones(mppi.n_dof)
        mppi.Cost.rest = torch.zeros(mppi.n_dof)
    if q_cur is not None:
        mppi.q_cur = torch.from_numpy(np.asarray(q_cur, dtype=np.float32))
    set_policy_state(mppi, K, rng, np.asarray(q0, np.float32), np.asarray(qf, np.float32), sigma_nom)

    fx = {
        "kind": np.array(kind), "N": N, "H": H, "dt": np.float32(dt), "k": k, "K": K,
        "obs": t2n(obs_t), "q0": t2n(q0_t), "qf": t2n(qf_t), "dh_params": t2n(dh),
        "dst_thr": np.float32(dst_thr), "ker_thr": np.float32(ker_thr),
        "ignored_links": np.asarray(mppi.ignored_links, dtype=np.int32),
        "p": p, "policy_upd_rate": np.float32(mppi.policy_upd_rate),
        "lin_thr": np.float32(mppi.DS.lin_thr),
        "cost_q_min": t2n(mppi.Cost.q_min), "cost_q_max": t2n(mppi.Cost.q_max),
        "n_iter": n_iter,
    }
    if extra:
        fx.update(extra)
    for it in range(n_iter):
        P = mppi.Policy
        pre = f"it{it}_"
        fx[pre + "q_cur"] = t2n(mppi.q_cur)
        fx[pre + "mu_c"] = t2n(P.mu_c[:K]); fx[pre + "sigma_c"] = t2n(P.sigma_c[:K]); fx[pre + "alpha_c"] = t2n(P.alpha_c[:K])
        P.sample_policy()
        fx[pre + "mu_tmp"] = t2n(P.mu_tmp[:, :K]); fx[pre + "sigma_tmp"] = t2n(P.sigma_tmp[:, :K]); fx[pre + "alpha_tmp"] = t2n(P.alpha_tmp[:, :K])
        with quiet():
            all_traj, dist_all, kval, dots, acts = mppi.propagate()
            cost = mppi.get_cost()
        fx[pre + "all_traj"] = t2n(all_traj); fx[pre + "closest_dist_all"] = t2n(dist_all)
        fx[pre + "kernel_val_all"] = t2n(kval); fx[pre + "dot_products"] = t2n(dots)
        fx[pre + "kernel_activations"] = t2n(acts)
        fx[pre + "qdot"] = t2n(mppi.qdot.reshape(N, -1))
        fx[pre + "norm_basis_n"] = t2n(mppi.norm_basis[:, :, :, 0])     # normal column only
        if N * H <= 1024:
            fx[pre + "norm_basis"] = t2n(mppi.norm_basis)               # full QR basis (for the "next" row)
        fx[pre + "cost"] = t2n(cost)
        # cost terms separately, to localise failures
        C = mppi.Cost
        fx[pre + "cost_goal"] = t2n(10 * C.goal_cost(all_traj[:, -1, :], C.qf))
        fx[pre + "cost_fk"] = t2n(10 * C.fk_cost(all_traj[:, -1, :]))
        beta = cost.mean() / 50
        w = torch.exp(-1 / beta * cost); w = w / w.sum()
        fx[pre + "w"] = t2n(w)
        fx[pre + "qdot_weighted"] = t2n(mppi.get_qdot("weighted"))
        fx[pre + "qdot_best"] = t2n(mppi.get_qdot("best"))
        # kernel-candidate search + kernel adding exactly as the planner does it (frankaPlanner.py:147-163),
        # on a COPY of the policy so that the scenario itself is not perturbed
        if it == 0:
            # thresholds at the medians of this scenario's own data, so that the candidate set is neither empty
            # nor everything (the drivers' -0.9 dot threshold selects almost nothing in such short runs)
            thr = (float(np.nanmedian(t2n(dist_all))), 0.5, float(np.nanmedian(t2n(dots))) if not np.isnan(t2n(dots)).all() else 0.0)
            cands = P.check_traj_for_kernels(all_traj, dist_all, dots, thr[0], thr[1], thr[2])
            fx["cand_thr"] = np.array(thr, np.float32)
            fx["cand_q"] = t2n(cands)
            if len(cands) > 0:
                import copy
                P2 = copy.deepcopy(P)
                norm, closest_idx = torch.norm(cands - mppi.q_cur, 2, -1).min(dim=0)
                idx_i, idx_h = torch.where((all_traj == cands[closest_idx]).all(dim=-1))
                with quiet():
                    P2.add_kernel(cands[closest_idx], dist_all[idx_i[0], idx_h[0]], mppi.norm_basis[idx_i[0], idx_h[0]].squeeze())
                fx["add_idx"] = np.array([int(closest_idx), int(idx_i[0]), int(idx_h[0])], np.int32)
                fx["add_n_kernels"] = int(P2.n_kernels)
                fx["add_mu_c"] = t2n(P2.mu_c[:P2.n_kernels]); fx["add_sigma_c"] = t2n(P2.sigma_c[:P2.n_kernels])
                fx["add_alpha_c"] = t2n(P2.alpha_c[:P2.n_kernels]); fx["add_gammas"] = t2n(P2.kernel_gammas[:P2.n_kernels])
                fx["add_basis"] = t2n(P2.kernel_obstacle_bases[:P2.n_kernels])
        with quiet():
            _, n_upd = mppi.shift_policy_means()
        fx[pre + "n_updated"] = int(n_upd)
        fx[pre + "mu_c_new"] = t2n(P.mu_c[:K]); fx[pre + "sigma_c_new"] = t2n(P.sigma_c[:K]); fx[pre + "alpha_c_new"] = t2n(P.alpha_c[:K])
        # per-stage intermediates at a late, diverse state (pass-1 matrix, top-k, vjp rows)
        if it == 0:
            h = max(0, H - 2)
            q = all_traj[:, h, :].clone()
            fx["st_q"] = t2n(q)
            nn_input = mppi.build_nn_input(q, mppi.obs)
            raw = mppi.nn_model.model_jit.forward(nn_input[:, 0:-1]).detach()
            d = raw / 100 if mppi.nn_model.out_channels == 9 else raw.clone()
            d = d - nn_input[:, -1].unsqueeze(1)
            d[:, mppi.ignored_links] = 1e6
            mind = d.min(1)[0].reshape(mppi.n_obs, N).transpose(0, 1)
            sd, si = mind.sort(dim=1)
            fx["st_mindist"] = t2n(mind); fx["st_sort_idx"] = t2n(si[:, :k]).astype(np.int32)
            fx["st_sort_dist"] = t2n(sd[:, :k])
            dist, grad = mppi.distance_repulsion_nn(q, aot=True)
            fx["st_distance"] = t2n(dist); fx["st_nn_grad"] = t2n(grad)
        # advance the robot like the drivers do (best rollout's first velocity)
        if advance == "best":
            mppi.q_cur = mppi.q_cur + mppi.get_qdot("best") * dt
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **fx)
    sz = os.path.getsize(path) / 1024
    ncoll = int((fx["it0_closest_dist_all"] < 0).sum())
    nact = int((fx["it0_kernel_activations"] > 0).sum())
    print(f"{name:28s} {sz:8.1f} KB  in-collision samples={ncoll:5d} active-kernel samples={nact:5d} "
          f"n_updated={fx['it0_n_updated']}")


def mlp_vectors(kind, nn_model, seed):
    """Raw MLP forward + vjp-of-argmin known answers on random rows (a1, a2 of SURVEY 8a)."""
    torch.manual_seed(seed)
    _, dof, out = MODELS[kind]
    B = 96
    x = torch.empty(B, dof + 3).uniform_(-2.5, 2.5)
    if kind.startswith("franka"):
        x[:, dof:] = torch.empty(B, 3).uniform_(-0.2, 1.0)
    else:
        x[:, dof:] = torch.empty(B, 3).uniform_(-7, 7); x[:, -1] = 0
    y = nn_model.model_jit.forward(x).detach()
    y2, g, mi = nn_model.functorch_vjp(x.clone())
    # pre-activations, to let tests flag rows whose ReLU mask is within rounding of flipping
    feats = torch.cat((x, torch.sin(x), torch.cos(x)), dim=-1)
    zmin = torch.full((B,), 1e9)
    hcur = feats
    mods = list(nn_model.model.layers)
    for gi, seq in enumerate(mods):
        if gi > 0:
            hcur = torch.cat((hcur, feats), dim=1)
        for li in range(len(seq) - (1 if gi == len(mods) - 1 else 0)):
            z = seq[li][0](hcur)
            zmin = torch.minimum(zmin, z.abs().min(dim=1)[0].detach())
            hcur = seq[li][1](z)
    np.savez_compressed(os.path.join(OUT, f"mlp_{kind}.npz"), x=t2n(x), y=t2n(y), y_vjp=t2n(y2), grad=t2n(g),
                        min_idx=t2n(mi).astype(np.int32), min_abs_preact=t2n(zmin))
    print(f"mlp_{kind}: y range [{float(y.min()):.3f}, {float(y.max()):.3f}]  min|z|={float(zmin.min()):.2e}")


def fk_vectors():
    """numeric_fk_model / numeric_fk_model_vec (fk_num.py:50-89): link sample points for the drivers' visualisation
    payloads (frankaPlanner.py:162-177) and the FK cost."""
    from fk_num import numeric_fk_model, numeric_fk_model_vec
    torch.manual_seed(5)
    out = {}
    for kind, n in (("franka", 7), ("planar2", 2)):
        dh, _ = robot_setup(kind)
        q = torch.empty(6, n).uniform_(-2.0, 2.0)
        links, pts_int = numeric_fk_model_vec(q, dh, 4)
        l1, p1 = numeric_fk_model(q[0], dh, 2)
        out.update({kind + "_q": t2n(q), kind + "_dh": t2n(dh), kind + "_links4": t2n(links), kind + "_int4": t2n(pts_int),
                    kind + "_links2_q0": t2n(l1), kind + "_int2_q0": t2n(p1)})
    np.savez_compressed(os.path.join(OUT, "fk_num.npz"), **out)
    print("fk_num:", {k: v.shape for k, v in out.items() if "links" in k})


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--only-fk" in sys.argv:
        return fk_vectors()
    if "--only-skip" in sys.argv:   # added after the other fixtures were committed: generate this network's files only
        m = load_model("franka_skip")
        export_weights("franka_skip", m)
        mlp_vectors("franka_skip", m, seed=11)
        run_scenario("franka_skip_shelf_K4", N=48, H=6, obs=scenes.shelf_scene(), k=5, K=4, seed=17, kind="franka_skip", nn_model=m,
                     dt=0.5, q0=scenes.FRANKA_Q0, qf=scenes.FRANKA_QF, dst_thr=0.01, ker_thr=0.1, alpha_s=3.0, sigma_nom=1.0)
        return
    fk_vectors()
    models = {k: load_model(k) for k in MODELS}
    for k, m in models.items():
        export_weights(k, m)
        mlp_vectors(k, m, seed=11)

    pi = math.pi
    # --- planar 2-DoF (BASELINE config 1 and the script's own shape) ---------------------------
    p2 = dict(kind="planar2", nn_model=models["planar2"], dt=0.3, q0=[-3.14, 0], qf=[3.14, 0], dst_thr=0.25,
              ker_thr=1e-3, alpha_s=2.0, sigma_nom=0.5, ignored_links=[])
    run_scenario("planar2_c1_K0", N=64, H=16, obs=scenes.planar2_scene(1), k=1, K=0, seed=1, **p2)
    run_scenario("planar2_c1_K3", N=64, H=16, obs=scenes.planar2_scene(1), k=1, K=3, seed=2, **p2)
    run_scenario("planar2_script_K2", N=100, H=10, obs=scenes.planar2_scene(2), k=2, K=2, seed=3, **p2)
    # --- planar 7-DoF (BASELINE config 2, reduced N/H) -----------------------------------------
    q0 = np.zeros(7, np.float32); q0[0] = pi / 2
    qf = np.zeros(7, np.float32); qf[0] = -pi / 2
    p7 = dict(kind="planar7", nn_model=models["planar7"], dt=0.3, q0=q0, qf=qf, dst_thr=0.25, ker_thr=1e-3,
              alpha_s=0.75, sigma_nom=0.5, ignored_links=[])
    run_scenario("planar7_K0", N=64, H=8, obs=scenes.planar7_scene(), k=1, K=0, seed=4, **p7)
    run_scenario("planar7_K4", N=64, H=8, obs=scenes.planar7_scene(), k=1, K=4, seed=5, **p7)
    run_scenario("planar7_128_K3", N=64, H=8, obs=scenes.planar7_scene(), k=2, K=3, seed=16,
                 **{**p7, "kind": "planar7_128", "nn_model": models["planar7_128"]})
    # --- Franka shelf (BASELINE config 3, reduced N/H) -----------------------------------------
    fr = dict(kind="franka", nn_model=models["franka"], dt=0.5, q0=scenes.FRANKA_Q0, qf=scenes.FRANKA_QF,
              dst_thr=0.01, ker_thr=0.1, alpha_s=3.0, sigma_nom=1.0)
    shelf = scenes.shelf_scene()
    run_scenario("franka_shelf_K0", N=48, H=6, obs=shelf, k=5, K=0, seed=6, **fr)
    run_scenario("franka_shelf_K6", N=48, H=6, obs=shelf, k=5, K=6, seed=7, n_iter=3, **fr)
    run_scenario("franka_sub40_K4", N=64, H=8, obs=shelf[::7][:40], k=5, K=4, seed=8, **fr)
    run_scenario("franka_cross_K3", N=40, H=10, obs=scenes.cross_scene(0.45), k=5, K=3, seed=9, **fr)
    # in-collision-heavy start: the robot reaching into the shelf
    q_in = np.array([0.0, 0.9, 0.0, -1.2, 0.0, 2.1, 0.0], np.float32)
    run_scenario("franka_shelf_collide_K4", N=48, H=6, obs=shelf, k=5, K=4, seed=10, q_cur=q_in, **fr)
    # integrator shape N=1, H=2 (alpha_s = 0), and K=50 edge, and at-goal NaN behaviour
    run_scenario("franka_integrator_N1", N=1, H=2, obs=shelf, k=5, K=5, seed=12,
                 **{**fr, "alpha_s": 0.0, "dt": 0.01, "dst_thr": 0.03})
    run_scenario("franka_sub40_K50", N=32, H=5, obs=shelf[::7][:40], k=5, K=50, seed=13, **fr)
    run_scenario("franka_at_goal_K2", N=16, H=4, obs=shelf[::7][:40], k=5, K=2, seed=14, q_cur=scenes.FRANKA_QF, **fr)
    # tanh 256x3 network (synthetic weights) on the shelf scene
    run_scenario("franka_tanh_shelf_K4", N=48, H=6, obs=shelf, k=5, K=4, seed=15, **{**fr, "kind": "franka_tanh",
                                                                                      "nn_model": models["franka_tanh"]})
    run_scenario("franka_skip_shelf_K4", N=48, H=6, obs=shelf, k=5, K=4, seed=17, **{**fr, "kind": "franka_skip",
                                                                                      "nn_model": models["franka_skip"]})


if __name__ == "__main__":
    main()

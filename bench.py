#!/usr/bin/env python3
"""Benchmark of the MPPI rollout + DS-modulation hot path on MI355X.

One "step" = one planner iteration of the reference (frankaPlanner.py:132-145):
sample_policy + propagate (H network evaluations over N x O pairs + modulation + Euler steps)
+ get_cost + shift_policy_means.  metric = modulated rollout-steps/s = N * H * steps / time,
the inverse of the reference's "Time per rollout step" (scripts/standalonePlanar2d.py:217).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 10 --warmup 2

Weak scaling: every rank (one process per GPU) owns the workload's rollouts; the only exchange is
the cost-weighted update: two tiny all-reduces issued by the library itself over RCCL on its own
stream (csrc/comm.hip).  torch.distributed (gloo) is used for the launcher plumbing only: shipping
the RCCL id, the barriers around the timed region and the max-over-ranks of the elapsed time.  If
the RCCL communicator cannot be created the run exits non-zero -- there is no fallback.
Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (weights, n_dof, C, scene, N, H, dt, k, dst_thr, ker_thr, alpha_s, sigma_nom, ignored)
    "franka_shelf_1024x32": dict(kind="franka", N=1024, H=32, dt=0.5, k=5, dst_thr=0.01, ker_thr=0.1, alpha_s=3.0,
                                 sigma=1.0, ignored=[0, 1, 2]),
    "franka_shelf_4096x32": dict(kind="franka", N=4096, H=32, dt=0.5, k=5, dst_thr=0.01, ker_thr=0.1, alpha_s=3.0,
                                 sigma=1.0, ignored=[0, 1, 2]),
    "planar7_1024x32": dict(kind="planar7", N=1024, H=32, dt=0.3, k=1, dst_thr=0.25, ker_thr=1e-3, alpha_s=0.75,
                            sigma=0.5, ignored=[]),
    # BASELINE configs[2] as BASELINE.json words it: "learned SDF MLP (256-256-256 tanh)", 4096 x 32 on the shelf.  The reference
    # ships no tanh weights: tests/golden/weights/franka_tanh.npz is its own MLPRegression(act_fn=Tanh) with seeded weights
    # (tools/make_golden.py) -- throughput does not depend on the values
    "franka_tanh_4096x32": dict(kind="franka_tanh", N=4096, H=32, dt=0.5, k=5, dst_thr=0.01, ker_thr=0.1, alpha_s=3.0,
                                sigma=1.0, ignored=[0, 1, 2], act="tanh"),
    # BASELINE configs[4] per GPU, as frankaPlanner.py:129-163 runs it: the obstacle set is replaced every iteration
    # (update_obstacles), kernel normals are re-evaluated (update_kernel_normal_bases), kernel candidates are searched on the
    # device and ONE kernel is added per iteration while candidates exist (Policy.add_kernel): every timed block starts from an
    # empty policy (Policy.reset_policy) at q0, so K grows from 0 inside the timed region and every iteration samples, evaluates
    # and reduces another kernel count
    "franka_dynamic_1024x32": dict(kind="franka", N=1024, H=32, dt=0.5, k=5, dst_thr=0.01, ker_thr=0.1, alpha_s=3.0,
                                   sigma=1.0, ignored=[0, 1, 2], dynamic=True),
    # BASELINE configs[3]'s shard: 32768 x 64 over 8 GPUs = 4096 x 64 per GPU
    "franka_shelf_4096x64": dict(kind="franka", N=4096, H=64, dt=0.5, k=5, dst_thr=0.01, ker_thr=0.1, alpha_s=3.0,
                                 sigma=1.0, ignored=[0, 1, 2]),
    # BASELINE configs[4]'s rollout count on ONE GPU (8 shards' worth): static shelf, fixed K
    "franka_shelf_8192x32": dict(kind="franka", N=8192, H=32, dt=0.5, k=5, dst_thr=0.01, ker_thr=0.1, alpha_s=3.0,
                                 sigma=1.0, ignored=[0, 1, 2]),
}
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz x 256 FLOP/clk
MFMA_F16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense fp16/bf16 MFMA (v_mfma_f32_32x32x16_f16), never the 2:1-sparsity figure
RCCL_FAILURE = [""]     # set when a rank could not form the library's RCCL communicator (run_workload)
PEAK_OF = {"k_pass1": ("f32 MFMA", MFMA_F32_PEAK_TFLOPS), "k_screen": ("f16 MFMA, f32 accumulate", MFMA_F16_PEAK_TFLOPS)}


def setup(wl, rank):
    from optimalmodulationds_amd import scenes
    w = WORKLOADS[wl]
    z = np.load(os.path.join(ROOT, "tests", "golden", "weights", w["kind"] + ".npz"))
    nl = len([k for k in z.files if k.startswith("W")])
    W = [z[f"W{i}"] for i in range(nl)]
    b = [z[f"b{i}"] for i in range(nl)]
    if w["kind"].startswith("franka"):
        obs, q0, qf, dh = scenes.shelf_scene(), scenes.FRANKA_Q0, scenes.FRANKA_QF, scenes.franka_dh_params()
        from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
        qmin, qmax = np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32)
    else:
        obs = scenes.planar7_scene()
        q0 = np.zeros(7, np.float32); q0[0] = np.pi / 2
        qf = np.zeros(7, np.float32); qf[0] = -np.pi / 2
        dh = scenes.planar_dh_params(7, 1.0)
        from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
        qmin, qmax = np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32)
    return w, W, b, obs, q0, qf, dh, qmin, qmax


def pmc_traffic(workload, kernel):
    """(HBM bytes per launch of the dominant kernel, the round whose profiles/ file holds it) from the committed rocprofv3 PMC
    summary of this workload (separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per MI355X_MICROARCH.md;
    tools/profile_refresh.sh).  The all-fp32 step's passes are filed under <workload>_fp32."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        key = workload + "_fp32" if (kernel == "k_pass1" and workload + "_fp32" in t) else workload
        return t[key][kernel]["traffic_bytes"], t.get("_round", "r03")
    except Exception:
        return None, None


def flops_per_row(W):
    """Algorithmic FLOPs of one network forward: 2 * sum(in*out) over the Linear layers."""
    return 2 * sum(int(w.shape[0]) * int(w.shape[1]) for w in W)


def cpu_baseline(w, W, b, obs, q0, qf, dh, qmin, qmax, K, H_full):
    """The reference's path as unfused torch-CPU ops (oracle/torch_baseline.py, pinned to the numpy oracle and through it
    to the reference's own outputs), timed on this box's host cores on a bounded sample of the same workload: 256 rollouts
    x 4 steps x all obstacles, full planner iteration (sample + propagate + cost + update), 3 warm-up + 5 timed iterations,
    median, at T = 8 threads (the setting of the survey's measurement of the reference itself) and
    at T = all physical cores.  propagate is linear in the horizon and the rest independent of it, so the full-horizon
    figure is N*H_full / (H_full/4 * t_propagate + t_rest)."""
    import platform
    import torch
    from oracle import omds_oracle as orc
    from oracle.torch_baseline import TorchPlanner, time_iterations
    m = orc.Mlp([x.astype(np.float32) for x in W], [x.astype(np.float32) for x in b])
    Ns, Hs = 256, 4
    prm = orc.Params(dst_thr=w["dst_thr"])
    pl = TorchPlanner(m, obs, qf, dh, qmin, qmax, dt=w["dt"], k=w["k"], ignored_links=w["ignored"], prm=prm)
    rng = np.random.RandomState(1234)
    n = q0.shape[0]
    s = (np.arange(K) + 0.5) / max(K, 1)
    mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, n))).astype(np.float32)
    sg_c = np.full(K, w["sigma"], np.float32)
    al_c = rng.standard_normal((K, n)).astype(np.float32)
    cpu = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        pass
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        phys = os.cpu_count() or 1
    out = {"unit": "rollout-steps/s", "kind": "port", "cpu": cpu,
           "sample": f"torch {torch.__version__} CPU, unfused op sequence of the reference, {Ns} rollouts x {Hs} steps x "
                     f"{obs.shape[0]} obstacles, K = {K}, full iteration, 3 warm-up + 5 timed, median; H = {H_full} figure = "
                     f"N*H / (H/{Hs} * t_propagate + t_rest)"}
    for label, T in (("t8", min(8, phys)), ("tall", phys)):
        tp, tr = time_iterations(pl, Ns, Hs, q0, mu_c, sg_c, al_c, w["alpha_s"], w["ker_thr"], T)
        out[label] = {"threads": int(T), "propagate_only": Ns * Hs / tp, "full_iteration": Ns * Hs / (tp + tr),
                      "full_iteration_at_workload_horizon": Ns * H_full / (H_full / Hs * tp + tr),
                      "t_propagate_s": tp, "t_rest_s": tr}
    # is the per-rollout rate flat in N on this box?  The workload's own rollout count, ONE horizon step, at 8 threads
    # (the extrapolation above scales a 256-rollout sample; this is the same path at N = 1024)
    Nw = int(w["N"]) if int(w["N"]) <= 1024 else 1024
    plw = TorchPlanner(m, obs, qf, dh, qmin, qmax, dt=w["dt"], k=w["k"], ignored_links=w["ignored"], prm=prm)
    tpw, trw = time_iterations(plw, Nw, 1, q0, mu_c, sg_c, al_c, w["alpha_s"], w["ker_thr"], min(8, phys), warmup=2, timed=3)
    out["at_workload_rollouts"] = {"threads": int(min(8, phys)), "rollouts": Nw, "steps": 1, "propagate_only": Nw / tpw, "t_propagate_s": tpw,
                                   "t_rest_s": trw, "note": "the same path at the workload's rollout count, one horizon step: compare propagate_only with t8's"}
    best = max(("t8", "tall"), key=lambda l: out[l]["full_iteration_at_workload_horizon"])   # more threads are not always faster here
    out["value"] = out[best]["full_iteration_at_workload_horizon"]
    out["cores"] = out[best]["threads"]
    return out


def measure(args, workload, steps, warmup, rank, world, local_rank, use_dist, dist, torch, time_fetch=False, prof=True, reps=1,
            screening=-1):
    """Times `reps` blocks of exactly `steps` planner iterations of `workload` on this rank's GPU (each block bracketed by a
    barrier + device synchronisation on both sides, elapsed = max over ranks); returns a dict of raw numbers.
    screening: -1 = the library's default path, 0 = the all-fp32 step (omds_set_screening(0))."""
    from optimalmodulationds_amd.dist import init_native_comm, sharded_update
    from optimalmodulationds_amd.engine import Engine
    w, W, b, obs, q0, qf, dh, qmin, qmax = setup(workload, rank)
    N, H, n, K = w["N"], w["H"], q0.shape[0], args.kernels
    eng = Engine(n, N, H, w["k"], max_obs=max(64, obs.shape[0]), device=local_rank)
    eng.set_mlp(W, b, act=w.get("act", "relu"))
    eng.set_obstacles(obs)
    if screening >= 0:
        eng.set_screening(screening)
    p = eng.params
    p.dt, p.dst_thr = w["dt"], w["dst_thr"]
    p.ignored_links = sum(1 << l for l in w["ignored"])
    eng.push_params()
    eng.set_ds(qf)
    eng.set_cost(dh, qmin, qmax)
    native = use_dist and (not args.share_gpu or args.try_rccl)
    if native:   # RCCL communicator owned by the library (csrc/comm.hip)
        err = ""
        try:
            init_native_comm(eng)
        except Exception as e:   # OmdsError(OMDS_ERR_RCCL): RCCL not loadable / communicator not formed on this rank
            err = f"rank {rank}: {e}"
        # the ranks must agree on the path: if any of them has no communicator, all of them exchange the same two small buffers
        # through the launcher's gloo group instead, and the JSON line says so (config.collectives) -- a labelled number, not a crash
        errs = [None] * world
        dist.all_gather_object(errs, err)
        if any(errs):
            native = False
            RCCL_FAILURE[0] = "; ".join(e for e in errs if e)[:400]
            if rank == 0:
                print("bench.py: RCCL communicator unavailable, host-mediated sums over gloo instead: " + RCCL_FAILURE[0], file=sys.stderr)
    # policy means: K kernel centres near the q0 -> qf segment (SURVEY 8d "policy state for timing")
    rng = np.random.RandomState(1234)
    s = (np.arange(K) + 0.5) / max(K, 1)
    mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, n))).astype(np.float32)
    sg_c = np.full(K, w["sigma"], np.float32)
    al_c = rng.standard_normal((K, n)).astype(np.float32)
    q_cur = q0.copy()
    dyn = bool(w.get("dynamic"))
    Kmax = eng.Kmax
    k_trace = []          # dynamic workload: K at the end of every timed block
    pick = np.random.RandomState(99)

    def reset_policy():   # Policy.reset_policy (policy.py:43-49) + the integrator back at q0: a timed block of the dynamic workload
        nonlocal mu_c, sg_c, al_c, q_cur, K
        K = 0
        mu_c, sg_c, al_c = np.zeros((Kmax, n), np.float32), np.zeros(Kmax, np.float32), np.zeros((Kmax, n), np.float32)
        q_cur = q0.copy()

    def choose_candidate(cand_q, cand_th):
        """frankaPlanner.py:149-161: the candidate closest to q_cur when it is within 0.1, else a random one; the rollout it came
        from is read back for closests_dist_all[i, h] and norm_basis[i, h] (its row only, omds_get_rollout_rows)."""
        d2 = np.linalg.norm(cand_q - q_cur, axis=1)
        j = int(np.argmin(d2)) if d2.min() < 1e-1 else int(pick.randint(cand_q.shape[0]))
        row = eng.get_rollout_rows([int(cand_th[j, 0])], want=("closest_dist_all", "normal"))
        _gamma, _normal = row["closest_dist_all"][0, cand_th[j, 1]], row["normal"][0, cand_th[j, 1]]   # kernel_gammas / kernel_obstacle_bases
        return cand_q[j].copy()

    def add_kernel(qk):
        """Policy.add_kernel (policy.py:129-151): centre = the candidate, sigma = nominal, alpha = the closest existing kernel's
        (0 for the first)."""
        nonlocal K
        if K < Kmax:
            mu_c[K], sg_c[K] = qk, w["sigma"]
            al_c[K] = al_c[int(np.argmin(np.linalg.norm(mu_c[:K] - qk, axis=1)))] if K else 0.0
            K += 1

    def iteration(it):
        nonlocal mu_c, sg_c, al_c, q_cur
        if dyn:
            # shelf translated by a slow sinusoid (the obstacle streamer's mechanism, obstacleStreamer.py:120-142),
            # then distance/normal at the K kernel centres (MPPI.update_kernel_normal_bases, MPPI.py:284-304)
            moved = obs.copy()
            moved[:, 1] += 0.05 * np.sin(0.3 * it)
            eng.set_obstacles(moved)
            if K:
                eng.dist_grad(mu_c[:K])
        eng.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, w["alpha_s"], K, seed=1234 * 1000003 + it, rollout_offset=rank * N)
        eng.propagate(q_cur)
        eng.cost(fetch=False)
        # two tiny all-reduces; the MINLOC gather for get_qdot('best') is not part of a planner iteration
        if use_dist and not native:   # --share-gpu: two ranks on ONE GPU cannot form a RCCL communicator; host-mediated gloo
            m2, s2, a2, mask, qd_w, _ = sharded_update(eng.cost_sum, eng.local_sums, K, n, H, 0.1, w["ker_thr"],
                                                       mu_c[:K], sg_c[:K], al_c[:K], want_best=False)
        else:                         # library path: RCCL on the context stream (single shard: the same kernels, no collective)
            m2, s2, a2, mask, qd_w, _, _ = eng.weighted_update_sharded(0.1, w["ker_thr"], mu_c, sg_c, al_c)
        if dyn:   # the means live in Kmax-row arrays whose first K rows are active
            mu_c[:K], sg_c[:K], al_c[:K] = m2, s2, a2
        else:
            mu_c, sg_c, al_c = m2, s2, a2
        q_cur = (q_cur + 0.1 * w["dt"] * qd_w).astype(np.float32)   # drift along the weighted rollout velocity: non-degenerate states
        if dyn:   # Policy.check_traj_for_kernels on the device (policy.py:153-175); only candidates cross PCIe; then add_kernel
            cq, cth, total = eng.kernel_candidates(0.03 - w["dst_thr"], 0.3, -0.9, mu_c, sg_c, K, cap=256)
            chosen = choose_candidate(cq, cth) if (cq.shape[0] and rank == 0) else None
            if use_dist:   # one planner decides (rank 0, whose shard holds the global rollout 0); every rank installs the same kernel,
                           # so the kernel count -- and with it the size of the update's all-reduce -- stays identical across ranks
                buf = torch.zeros(1 + n, dtype=torch.float32)
                if chosen is not None:
                    buf[0] = 1.0
                    buf[1:] = torch.from_numpy(chosen)
                dist.broadcast(buf, 0)
                chosen = buf[1:].numpy().copy() if float(buf[0]) > 0 else None
            if chosen is not None:
                add_kernel(chosen)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if dyn:
        reset_policy()
    for it in range(warmup):
        iteration(it)
    # HIP events around every launch of the dominant kernel (keeps a batch on ONE stream).  An event record between two
    # kernels idles the GPU for ~6 us (tools/gap_probe.py); sampling every 8th launch instead (prof_enable(8)) was measured:
    # same iteration time within noise (back-to-back launches run ~0.7 % slower each), so every launch is timed
    eng.prof_enable(args.prof_stride if prof else 0)
    eng.prof_reset()
    els = []
    for rep in range(reps):
        if dyn:
            reset_policy()
        barrier()
        t0 = time.perf_counter()
        for it in range(steps):
            iteration(warmup + rep * steps + it)
        eng.lib.omds_sync(eng.h)
        barrier()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        els.append(el)
        k_trace.append(int(K))
    el = float(np.median(els))
    p1_ms, p1_launches, p1_rows = eng.prof_read()
    _, _, p1_flops, p1_kernel = eng.prof_read_ex()
    fetch_ms = None
    if time_fetch:
        eng.get_rollouts()
        tf = time.perf_counter()
        for _ in range(5):
            eng.get_rollouts()
        fetch_ms = (time.perf_counter() - tf) / 5 * 1e3

    scr = eng.screen_stats()
    eng.close()
    return dict(els=els, scr=scr, k_trace=k_trace, dh=dh, qmin=qmin, qmax=qmax, w=w, W=W, b=b, obs=obs, q0=q0, qf=qf, N=N, H=H, K=K, el=el, p1_ms=p1_ms, p1_launches=p1_launches,
                p1_rows=p1_rows, p1_flops=p1_flops, p1_kernel=p1_kernel, fetch_ms=fetch_ms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="franka_shelf_1024x32", choices=sorted(WORKLOADS))
    ap.add_argument("--kernels", type=int, default=10, help="active RBF navigation kernels K")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the all-fp32 figure and the other workloads reported under 'also'")
    ap.add_argument("--try-rccl", action="store_true", help="with --share-gpu: attempt the RCCL communicator anyway (test of the failure path with OMDS_RCCL_LIB)")
    ap.add_argument("--reps", type=int, default=10, help="timed blocks of --steps iterations each; value = median block")
    ap.add_argument("--prof-stride", type=int, default=8,
                    help="HIP events around every n-th launch of the dominant kernel inside the timed region (an event record between two "
                         "launches idles the GPU for a few us: every launch costs 2.4 %% of value at 190 us per step, every 8th 0.2 %%)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test mode for a 1-GPU box: all ranks use GPU 0 and exchange through the host (gloo); never a scaling number")
    ap.add_argument("--screening", type=int, default=-1, choices=(-1, 0, 1),
                    help="-1: the library's own choice (default); 0: the all-fp32 step as the primary measurement (omds_set_screening(0)); 1: forced on")
    ap.add_argument("--time-fetch", action="store_true", help="also report the cost of fetching all rollout tensors to the host")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on this driver
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # under torchrun, also at N=1
    if use_dist:
        # launcher plumbing only (RCCL id broadcast, barriers, max over ranks of the elapsed time); the data path's
        # collectives are the library's own RCCL calls (csrc/comm.hip)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    r = measure(args, args.workload, args.steps, args.warmup, rank, world, local_rank, use_dist, dist, torch, args.time_fetch,
                reps=max(1, args.reps), screening=args.screening)
    w, W, b, obs, q0, qf, N, H, K, el = (r[k] for k in ("w", "W", "b", "obs", "q0", "qf", "N", "H", "K", "el"))
    p1_ms, p1_launches, p1_rows, fetch_ms = r["p1_ms"], r["p1_launches"], r["p1_rows"], r["fetch_ms"]

    def roofline(rr, workload):
        ach = rr["p1_flops"] / (rr["p1_ms"] * 1e-3) / 1e12 if rr["p1_ms"] > 0 else 0.0
        pipe, peak = PEAK_OF.get(rr["p1_kernel"], PEAK_OF["k_pass1"])
        traffic, traffic_round = pmc_traffic(workload, rr["p1_kernel"])
        return {"bound": "mfma", "kernel": rr["p1_kernel"], "pipe": pipe, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": traffic,
                "traffic_source": f"profiles/pmc_traffic.json = profiles/{traffic_round}_pmc_hbm*.txt (rocprofv3 --pmc passes of this workload, "
                                  f"collected in round {traffic_round} by tools/profile_refresh.sh and committed; not re-collected by this run)",
                "launches": int(rr["p1_launches"]), "launch_sampling": f"HIP events around every {args.prof_stride}-th launch inside the timed blocks",
                "avg_launch_ms": rr["p1_ms"] / max(rr["p1_launches"], 1),
                "flops_per_launch": rr["p1_flops"] / max(rr["p1_launches"], 1),
                "profile": "profiles/r04_kernel_trace_stats%s.txt (rocprofv3 --kernel-trace --stats of this command%s)" %
                           (("", "") if rr["p1_kernel"] != "k_pass1" else ("_fp32", " with the library's screening switched off")),
                **({"flops_basis": "algorithmic: N x O pairs x the dense network.  k_screen does not issue the MFMAs of a k-chunk whose 16 hidden "
                                   "units are zero for all 32 pairs of a wave (exact: it adds nothing); the library sorts the units by how often "
                                   "they fire so that the silent ones of a trained ReLU network fill whole chunks (screening.unit_reorders, "
                                   "screening.units_never_fired; SQ_INSTS_MFMA in profiles/r04_pmc_sq.txt is what the matrix pipe executed)"}
                   if rr["p1_kernel"] == "k_screen" and rr["scr"].get("unit_reorders", 0) > 0 else {})}

    def rate(rr, steps):
        v = [world * rr["N"] * rr["H"] * steps / e for e in rr["els"]]
        return {"value": float(np.median(v)), "value_min": float(min(v)), "value_max": float(max(v)), "reps": len(v),
                "ms_per_step": 1e3 * float(np.median(rr["els"])) / steps,
                "rep_ms_per_step": [round(1e3 * e / steps, 4) for e in rr["els"]]}

    fp32 = None
    also = None
    if not args.no_secondary:
        # the same iterations with screening off: every pass-1 row in fp32 (k_pass1 + k_tail), the arithmetic of the reference
        r32 = measure(args, args.workload, args.steps, 1, rank, world, local_rank, use_dist, dist, torch, reps=3, screening=0)
        fp32 = dict(rate(r32, args.steps), roofline=roofline(r32, args.workload))
        also = []
        for wl2, st2, rp2 in (("planar7_1024x32", 10, 5), ("franka_shelf_4096x32", 5, 3), ("franka_dynamic_1024x32", 20, 3),
                              ("franka_tanh_4096x32", 5, 3), ("franka_shelf_4096x64", 3, 3), ("franka_shelf_8192x32", 3, 3)):
            if wl2 == args.workload:
                continue
            r2 = measure(args, wl2, st2, 1, rank, world, local_rank, use_dist, dist, torch, prof=False, reps=rp2)
            e2 = dict({"workload": wl2, "unit": "rollout-steps/s", "steps": st2}, **rate(r2, st2))
            if WORKLOADS[wl2].get("dynamic"):
                e2["kernels_at_block_end"] = r2["k_trace"]   # K grows from 0 inside every timed block (one add_kernel per iteration at most)
            e2["screening"] = {k2: r2["scr"][k2] for k2 in ("active", "eps", "candidates_per_rollout_step", "fallbacks", "audit_max_err",
                                                          "audit_rows_per_rollout_step", "calibrations", "suspended", "sweeps", "sweep_max_err")}
            also.append(e2)
    if rank == 0:
        act = w.get("act", "relu")
        out = {
            "metric": "modulated rollout-steps/sec", **{k2: v2 for k2, v2 in rate(r, args.steps).items() if k2 not in ("ms_per_step", "rep_ms_per_step")},
            "unit": "rollout-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rate(r, args.steps)["ms_per_step"], "rep_ms_per_step": rate(r, args.steps)["rep_ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # dtype = what every number the step RETURNS is computed in.  With screening active the N*O first-pass evaluations
            # that only feed the obstacle selection run in screen_dtype (fp16 inputs, fp32 accumulate); value_fp32_only is the
            # same run with every row in fp32
            "dtype": "f32", "screen_dtype": "f16" if r["scr"]["active"] else None, "data": "synthetic",
            "config": {"workload": args.workload,
                       # what the arithmetic is, in one sentence, where the driver's parser keeps it
                       "precision": ("fp32 outputs (every distance, gradient, velocity and cost the step returns comes from fp32 kernels); the N x O "
                                     "first-pass evaluations that only feed the obstacle selection are screened in f16 and the candidates re-evaluated "
                                     "in fp32 -- identity with the all-fp32 step is conditional on a measured bound eps (profiles/r04_screen_error_hist.txt: "
                                     "0 of > 1e11 unevaluated pairs above eps / 2); value_fp32_only / roofline.fp32_only = the same iterations with "
                                     "every row in fp32") if r["scr"]["active"] else "fp32 throughout",
                       "rollouts_per_gpu": N, "horizon": H, "obstacles": int(obs.shape[0]),
                       "n_closest": w["k"], "active_kernels": K, "network": "x".join(str(x.shape[1]) for x in W) + f"x{W[-1].shape[0]} {act} "
                       + ("(shipped reference weights)" if act == "relu" else "(seeded synthetic weights)"),
                       "parallelism": f"rollout-sharded x{world}", "collectives": (("gloo-host (RCCL unavailable: " + RCCL_FAILURE[0] + ")" if RCCL_FAILURE[0] else ("gloo-host (--share-gpu test mode)" if args.share_gpu else "rccl")) if use_dist else "none")},
            "roofline": roofline(r, args.workload),
        }
        if fp32 is not None:
            out["value_fp32_only"] = fp32["value"]
            out["fp32_only"] = fp32
            # the precision-matched figure where the driver's parser keeps it: the all-fp32 iteration and its dominant kernel
            rf = fp32["roofline"]
            out["roofline"]["fp32_only"] = {"kernel": rf["kernel"], "pipe": rf["pipe"], "frac": rf["frac"], "achieved": rf["achieved"], "peak": rf["peak"],
                                            "avg_launch_ms": rf["avg_launch_ms"], "launches": rf["launches"], "value": fp32["value"],
                                            "ms_per_step": fp32["ms_per_step"], "profile": rf["profile"]}
        out["screening"] = r["scr"]   # fp16 screening of pass 1 + exact fp32 re-selection + audit sample; inactive = fp32 pass 1
        if fetch_ms is not None:
            out["fetch_all_rollouts_ms"] = fetch_ms
        if also is not None:
            out["also"] = also
        if not args.no_cpu_baseline and world == 1 and act == "relu":   # oracle/torch_baseline.py restates the shipped (ReLU) path
            out["cpu_baseline"] = cpu_baseline(w, W, b, obs, q0, qf, r["dh"], r["qmin"], r["qmax"], K, H)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

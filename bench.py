#!/usr/bin/env python3
"""Benchmark of the MPPI rollout + DS-modulation hot path on MI355X.

One "step" = one planner iteration of the reference (frankaPlanner.py:132-145):
sample_policy + propagate (H network evaluations over N x O pairs + modulation + Euler steps)
+ get_cost + shift_policy_means.  metric = modulated rollout-steps/s = N * H * steps / time,
the inverse of the reference's "Time per rollout step" (scripts/standalonePlanar2d.py:217).

    python bench.py                                   # 1 GPU
    python bench.py --gpus 8                          # starts 8 ranks itself (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 10 --warmup 2      # the driver's form: the same 8 ranks

What `value` is.  The PRIMARY measurement is the all-fp32 step (--path fp32, `dtype: "f32"`), the library's DEFAULT: every network
evaluation in the reference's own arithmetic -- ascending-k fmaf chains from zero, bias last, bit for bit the oracle's -- (k_pass1 +
k_tail), priced against the fp32 MFMA peak in `roofline`.  The OPT-IN screened step (omds_set_screening(ctx, 2, 0)) evaluates
the N x O first-pass rows -- which only feed the sort that picks the k closest obstacles -- in f16 and re-evaluates the candidates in
fp32; every number it returns is an fp32 number and bit-identical to the all-fp32 step's as long as a measured error bound holds
(include/omds.h).  That step is timed in the same run and reported as `value_screened` with its own roofline block
(`roofline.screened`: k_screen against the dense f16 MFMA peak, algorithmic `frac` and executed `frac_issued`).

Ranks.  One process per GPU.  `--gpus N` without a launcher environment starts the N ranks itself: a child
`python -m torch.distributed.run` is spawned BEFORE this process makes any GPU call, rank 0's JSON line is relayed, and the exit
status is non-zero when any rank fails, when the box has fewer than N GPUs, or when the RCCL communicator cannot be formed.  There
is no silent single-GPU run and no silent host-mediated run: the gloo form of the exchange exists only behind
--allow-host-collectives (and in --share-gpu, a test mode for 1-GPU boxes) and labels the line.
--scaling weak (default): every rank owns the workload's N rollouts; --scaling strong: the workload's N rollouts are split over
the ranks (the metric is quoted on 1024 x 32 at 1/2/4/8 GPUs).  The only exchange is the cost-weighted update: two tiny
all-reduces issued by the library itself over RCCL on its own stream (csrc/comm.hip).  torch.distributed (gloo) carries the
launcher plumbing only: the RCCL id, the barriers around the timed region, the max over ranks of the elapsed time.  The GPU is
touched through the C-ABI alone (device count, contexts, synchronisation): PyTorch never initialises the device in this process.
Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
ROUND = "r06"      # the profiles/ files of this round (tools/profile_refresh.sh)

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
PEAK_OF = {"k_pass1": ("f32 MFMA", MFMA_F32_PEAK_TFLOPS), "k_screen": ("f16 MFMA, f32 accumulate", MFMA_F16_PEAK_TFLOPS)}
FLOP_PER_MFMA = {"k_screen": 2 * 32 * 32 * 16}    # v_mfma_f32_32x32x16_f16


class Launch:
    """Where this process sits among the ranks, and the launcher's gloo group (None at one rank without a launcher)."""
    def __init__(self, rank=0, world=1, local_rank=0, dist=None, torch=None):
        self.rank, self.world, self.local_rank, self.dist, self.torch = rank, world, local_rank, dist, torch
        self.collectives = "none"       # "rccl" | "gloo-host (...)" once a multi-rank exchange exists
        self.rccl = None                # what the library's communicator reported on every rank (measure())

    @property
    def use_dist(self):
        return self.dist is not None


# ---- self-launch -----------------------------------------------------------------------------------------------------------
def physical_cores():
    """Physical cores of this box: distinct (physical id, core id) pairs of /proc/cpuinfo (no psutil needed); logical CPUs when
    the file has no topology."""
    try:
        pairs, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            pairs.add((phys, core))
        if pairs:
            return min(len(pairs), len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def visible_devices():
    """HIP devices of this box, counted in a short-lived child through the C-ABI (omds_device_count): the launcher process itself
    never touches the GPU."""
    import subprocess
    code = "import sys; sys.path.insert(0, %r); from optimalmodulationds_amd import _lib; print('NDEV', _lib.device_count())" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    for line in r.stdout.splitlines():
        if line.startswith("NDEV "):
            return int(line.split()[1])
    raise SystemExit("bench.py: could not count the GPUs through libomds_hip.so (there is no CPU fallback):\n" + r.stdout[-1000:] + r.stderr[-2000:])


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher environment: start the N ranks as a child `torch.distributed.run` (never an
    exec, and before this process has made any GPU call), relay rank 0's JSON line, exit non-zero unless every rank succeeded."""
    import socket
    import subprocess
    ndev = visible_devices()
    need = 1 if args.share_gpu else args.gpus
    if ndev < need:
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {need} GPU(s), this box has {ndev}: refusing to run "
                         f"(a {args.gpus}-GPU line is never produced on fewer devices; --share-gpu is the labelled 1-GPU test mode)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)      # stderr passes through
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    if r.returncode != 0 or len(lines) != 1:
        sys.stderr.write(r.stdout[-4000:])
        raise SystemExit(f"bench.py: the {args.gpus}-rank run failed (launcher exit {r.returncode}, {len(lines)} result lines): no number is reported")
    if json.loads(lines[0]).get("n_gpus") != args.gpus:
        raise SystemExit("bench.py: the ranks reported another world size than --gpus")
    print(lines[0])
    sys.exit(0)


# ---- workload ---------------------------------------------------------------------------------------------------------------
def setup(wl):
    from optimalmodulationds_amd import scenes
    w = WORKLOADS[wl]
    z = np.load(os.path.join(ROOT, "tests", "golden", "weights", w["kind"] + ".npz"))
    nl = len([k for k in z.files if k.startswith("W")])
    W = [z[f"W{i}"] for i in range(nl)]
    b = [z[f"b{i}"] for i in range(nl)]
    from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
    qmin, qmax = np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32)
    if w["kind"].startswith("franka"):
        obs, q0, qf, dh = scenes.shelf_scene(), scenes.FRANKA_Q0, scenes.FRANKA_QF, scenes.franka_dh_params()
    else:
        obs = scenes.planar7_scene()
        q0 = np.zeros(7, np.float32); q0[0] = np.pi / 2
        qf = np.zeros(7, np.float32); qf[0] = -np.pi / 2
        dh = scenes.planar_dh_params(7, 1.0)
    return w, W, b, obs, q0, qf, dh, qmin, qmax


def pmc_profile(workload, kernel, fp32):
    """(HBM bytes per launch, SQ_INSTS_MFMA per launch, round) of `kernel` from the committed rocprofv3 PMC summaries of this
    workload (separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md; tools/profile_refresh.sh).  The all-fp32 step's
    passes are filed under <workload>_fp32."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        key = workload + "_fp32" if (fp32 and workload + "_fp32" in t) else workload
        e = t[key][kernel]
        return e.get("traffic_bytes"), e.get("mfma_insts"), t.get("_round", "r04")
    except Exception:
        return None, None, None


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
    phys = physical_cores()
    out = {"unit": "rollout-steps/s", "kind": "port", "cpu": cpu, "physical_cores": int(phys), "logical_cpus": int(os.cpu_count() or 1),
           "sample": f"torch {torch.__version__} CPU, unfused op sequence of the reference, {Ns} rollouts x {Hs} steps x "
                     f"{obs.shape[0]} obstacles, K = {K}, full iteration, 3 warm-up + 5 timed, median; H = {H_full} figure = "
                     f"N*H / (H/{Hs} * t_propagate + t_rest).  KINDER to the CPU than the reference is: the port evaluates the FK cost of all "
                     f"rollouts in one vectorised call, the reference runs a TorchScript loop over the rollouts (fk_num.py:87-88, ~2.4 ms per rollout: "
                     f"~2.5 s per iteration at 1024 rollouts, more than the whole ported iteration)"}
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


def measure(args, L, workload, steps, warmup, time_fetch=False, prof=True, reps=1, screening=0, kernels=None, rollouts=None):
    """Times `reps` blocks of exactly `steps` planner iterations of `workload` on this rank's GPU (each block bracketed by a
    barrier + device synchronisation on both sides, elapsed = max over ranks); returns a dict of raw numbers.
    screening: 0 = the all-fp32 step (the library's default; set explicitly), 2 = the opt-in screened step where it pays (omds_set_screening(ctx, 2, 0)),
    1 = screening forced on.
    rollouts: this rank's rollout count when it is not the workload's (strong scaling, the shard sweep)."""
    from optimalmodulationds_amd.engine import Engine
    rank, world, dist, torch = L.rank, L.world, L.dist, L.torch
    w, W, b, obs, q0, qf, dh, qmin, qmax = setup(workload)
    N = int(rollouts) if rollouts else w["N"]
    H, n = w["H"], q0.shape[0]
    K = args.kernels if kernels is None else int(kernels)
    eng = Engine(n, N, H, w["k"], max_obs=max(64, obs.shape[0]), device=L.local_rank, flags=(8 if getattr(args, "dense_pass1", False) else 0))
    eng.set_mlp(W, b, act=w.get("act", "relu"))
    eng.set_obstacles(obs)
    eng.set_screening(screening)
    p = eng.params
    p.dt, p.dst_thr = w["dt"], w["dst_thr"]
    p.ignored_links = sum(1 << l for l in w["ignored"])
    eng.push_params()
    eng.set_ds(qf)
    eng.set_cost(dh, qmin, qmax)
    native = False
    if L.use_dist:
        from optimalmodulationds_amd.dist import init_native_comm, sharded_update
        err = ""
        if not args.share_gpu or args.try_rccl:   # RCCL communicator owned by the library (csrc/comm.hip)
            try:
                init_native_comm(eng)   # every rank probes the loader and the ranks agree before the collective init (dist.py)
            except Exception as e:      # OmdsError(OMDS_ERR_RCCL): RCCL not loadable / communicator not formed
                err = f"rank {rank}: {e}"
            errs = [None] * world
            dist.all_gather_object(errs, err)
            err = "; ".join(e for e in errs if e)[:600]
            native = not err
        else:
            err = "--share-gpu test mode"
        if native:
            L.collectives = "rccl"
            # self-certification: the world size the LIBRARY's RCCL communicator reports on every rank and the devices each
            # rank's process sees through the C-ABI, gathered over gloo and required to agree
            from optimalmodulationds_amd import _lib as _olib
            mine = {"rank": rank, "rccl_rank": int(eng.comm_info()[0]), "rccl_world": int(eng.comm_info()[1]), "devices_seen": int(_olib.device_count()),
                    "device": int(L.local_rank)}
            allr = [None] * world
            dist.all_gather_object(allr, mine)
            worlds = sorted({a["rccl_world"] for a in allr})
            if worlds != [world] or sorted(a["rccl_rank"] for a in allr) != list(range(world)):
                eng.close()
                if rank == 0:
                    print(f"bench.py: the RCCL communicator does not span the {world} ranks: {allr}", file=sys.stderr)
                dist.destroy_process_group()
                sys.exit(3)
            L.rccl = {"rccl_ranks": world, "rccl_rank_of_each_process": [a["rccl_rank"] for a in allr],
                      "devices_seen": [a["devices_seen"] for a in allr], "device_of_each_process": [a["device"] for a in allr]}
        elif args.allow_host_collectives or err == "--share-gpu test mode":
            # asked for: the same two small buffers through the launcher's gloo group, and the line says so
            L.collectives = "gloo-host (--share-gpu test mode)" if err == "--share-gpu test mode" else "gloo-host (RCCL unavailable: " + err + ")"
            if rank == 0 and err != "--share-gpu test mode":
                print("bench.py: RCCL communicator unavailable, host-mediated sums over gloo instead (--allow-host-collectives): " + err, file=sys.stderr)
        else:
            eng.close()
            if rank == 0:
                print("bench.py: the RCCL communicator could not be formed: " + err + "\nbench.py: refusing to produce a "
                      f"{world}-GPU number over host-mediated sums (--allow-host-collectives runs that form, labelled)", file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(3)
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
        if L.use_dist and not native:   # host-mediated form (asked for): the launcher's gloo group
            m2, s2, a2, mask, qd_w, _ = sharded_update(eng.cost_sum, eng.local_sums, K, n, H, 0.1, w["ker_thr"],
                                                       mu_c[:K], sg_c[:K], al_c[:K], want_best=False)
        else:                           # library path: RCCL on the context stream (single shard: the same kernels, no collective)
            m2, s2, a2, mask, qd_w, _, _ = eng.weighted_update_sharded(0.1, w["ker_thr"], mu_c, sg_c, al_c)
        if dyn:   # the means live in Kmax-row arrays whose first K rows are active
            mu_c[:K], sg_c[:K], al_c[:K] = m2, s2, a2
        else:
            mu_c, sg_c, al_c = m2, s2, a2
        q_cur = (q_cur + 0.1 * w["dt"] * qd_w).astype(np.float32)   # drift along the weighted rollout velocity: non-degenerate states
        if dyn:   # Policy.check_traj_for_kernels on the device (policy.py:153-175); only candidates cross PCIe; then add_kernel
            cq, cth, total = eng.kernel_candidates(0.03 - w["dst_thr"], 0.3, -0.9, mu_c, sg_c, K, cap=256)
            chosen = choose_candidate(cq, cth) if (cq.shape[0] and rank == 0) else None
            if L.use_dist:   # one planner decides (rank 0, whose shard holds the global rollout 0); every rank installs the same kernel,
                             # so the kernel count -- and with it the size of the update's all-reduce -- stays identical across ranks
                buf = torch.zeros(1 + n, dtype=torch.float32)
                if chosen is not None:
                    buf[0] = 1.0
                    buf[1:] = torch.from_numpy(chosen)
                dist.broadcast(buf, 0)
                chosen = buf[1:].numpy().copy() if float(buf[0]) > 0 else None
            if chosen is not None:
                add_kernel(chosen)

    def barrier():     # launcher barrier + the device idle (omds_sync: hipStreamSynchronize of the context's stream -- all of this
        if L.use_dist:   # rank's GPU work is on it; PyTorch holds no device context in this process)
            dist.barrier()
        eng.sync()

    if dyn:
        reset_policy()
    for it in range(warmup):
        iteration(it)
    # HIP events around every n-th launch of the dominant kernel, on the context's stream.  An event record between two
    # kernels idles the GPU for ~6 us (tools/gap_probe.py), so throughput runs sample (--prof-stride)
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
        barrier()
        el = time.perf_counter() - t0
        if L.use_dist:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        els.append(el)
        k_trace.append(int(K))
    p1_ms, p1_launches, p1_rows = eng.prof_read()
    _, _, p1_flops, p1_kernel = eng.prof_read_ex()
    fetch_ms = None
    if time_fetch:     # omds_get_rollouts: all rollout tensors to the host in the reference's layout (never part of `value`)
        eng.get_rollouts()
        tf = time.perf_counter()
        for _ in range(5):
            eng.get_rollouts()
        fetch_ms = (time.perf_counter() - tf) / 5 * 1e3
    scr = eng.screen_stats()
    skip = eng.pass1_skip_stats()
    eng.close()
    return dict(els=els, scr=scr, skip=skip, k_trace=k_trace, dh=dh, qmin=qmin, qmax=qmax, w=w, W=W, b=b, obs=obs, q0=q0, qf=qf, N=N, H=H, K=K,
                p1_ms=p1_ms, p1_launches=p1_launches, p1_rows=p1_rows, p1_flops=p1_flops, p1_kernel=p1_kernel, fetch_ms=fetch_ms,
                steps=steps)


def rate(rr, n_total):
    """value / min / max / per-block times of one measure() result; n_total = rollouts over all ranks."""
    v = [n_total * rr["H"] * rr["steps"] / e for e in rr["els"]]
    return {"value": float(np.median(v)), "value_min": float(min(v)), "value_max": float(max(v)), "reps": len(v),
            "ms_per_step": 1e3 * float(np.median(rr["els"])) / rr["steps"],
            "rep_ms_per_step": [round(1e3 * e / rr["steps"], 4) for e in rr["els"]]}


def roofline(args, rr, workload, fp32):
    """The dominant kernel of the measured step against its MFMA peak.  `achieved` / `frac` count ALGORITHMIC FLOPs (SURVEY 8d:
    N x O pairs x the dense network's 2 * sum(in * out)) over the HIP-event launch time; `frac_issued` counts the FLOPs of the
    MFMA instructions the kernel EXECUTES: k_pass1 pads layer 1's 3(n+3) inputs to K = 32 and the last layer to 16 columns -- an
    analytic count; k_screen skips the k-chunks whose 16 hidden units are zero for all
    32 pairs of a wave -- data-dependent, so SQ_INSTS_MFMA of the committed PMC pass x 32768 FLOP."""
    ach = rr["p1_flops"] / (rr["p1_ms"] * 1e-3) / 1e12 if rr["p1_ms"] > 0 else 0.0
    kern = rr["p1_kernel"]
    pipe, peak = PEAK_OF.get(kern, PEAK_OF["k_pass1"])
    traffic, insts, prof_round = pmc_profile(workload, kern, fp32)
    launches = max(int(rr["p1_launches"]), 1)
    avg_ms = rr["p1_ms"] / launches
    flops_launch = rr["p1_flops"] / launches
    out = {"bound": "mfma", "kernel": kern, "pipe": pipe, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak}
    W = rr["W"]
    f_alg = 2 * sum(int(x.shape[0]) * int(x.shape[1]) for x in W)
    if kern == "k_pass1":      # layer 1 over the encoded inputs padded to K = 32, hidden layers as they are, the last layer padded to 16 columns
        f_iss = 2 * 32 * int(W[0].shape[0]) + 2 * sum(int(x.shape[0]) * int(x.shape[1]) for x in W[1:-1]) + 2 * int(W[-1].shape[1]) * 16
        basis = (f"analytic: {f_iss} FLOP of MFMA per pair (layer 1 as a K = 32 product over the 3(n+3) encoded inputs -- the reference's single "
                 f"chain does not split into two precomputed halves --, hidden layers, last layer padded to 16 columns) against {f_alg} algorithmic")
        skip = rr.get("skip") or {}
        if skip.get("active") and skip.get("tiles", 0) > 0 and len(skip.get("chunks", [])) >= len(W) - 1:
            # the exact zero-skip (omds_pass1_skip_stats): every tile stores a hidden level compacted to the units that fire in it and
            # the level's consumer multiplies ceil(T / 8) k-chunks (the last layer: ceil(T / 16)) instead of 32 (16)
            ch, un = skip["chunks"][:len(W) - 1], skip["units"][:len(W) - 1]
            f_iss = 2 * 32 * int(W[0].shape[0]) + 2 * sum(8 * ch[l] * int(W[l + 1].shape[0]) for l in range(len(W) - 2)) + 2 * 16 * ch[-1] * 16
            basis = (f"analytic: {f_iss:.0f} FLOP of MFMA per pair executed -- layer 1 as a K = 32 product, then the EXACT ZERO-SKIP: a tile keeps only the hidden "
                     f"units that fire in it ({', '.join(f'{u:.0f}' for u in un)} of 256 per level on average) and their consumers multiply "
                     f"{', '.join(f'{c:.1f}' for c in ch[:-1])} of 32 k-chunks and {ch[-1]:.1f} of 16 in the last layer (fmaf(0, w, acc) = acc: same bits, include/omds.h "
                     f"omds_pass1_skip_stats) -- against {f_alg} algorithmic: `frac` credits the dense network's work and exceeds 1, `frac_issued` is what "
                     f"the matrix pipe executed")
            out["zero_skip"] = {"mean_units_per_tile": [round(u, 1) for u in un], "mean_chunks": [round(c, 2) for c in ch], "tiles": skip["tiles"]}
        out["frac_issued"] = out["frac"] * f_iss / f_alg
        out["frac_issued_basis"] = basis + (f"; SQ_INSTS_MFMA of the committed pass: {insts:.0f} per launch" if insts else "")
    elif kern in FLOP_PER_MFMA and insts and avg_ms > 0:
        out["frac_issued"] = insts * FLOP_PER_MFMA[kern] / (avg_ms * 1e-3) / 1e12 / peak
        out["mfma_insts_per_launch"] = insts
        out["frac_issued_basis"] = (f"SQ_INSTS_MFMA = {insts:.0f} per launch (profiles/{prof_round}_pmc_sq.txt, the same command under rocprofv3 --pmc) x "
                                    f"{FLOP_PER_MFMA[kern]} FLOP over this run's launch time; the algorithmic FLOPs of a launch are "
                                    f"{flops_launch / FLOP_PER_MFMA[kern]:.0f} instruction-equivalents: the kernel does not multiply k-chunks of 16 hidden "
                                    "units that are zero for all 32 pairs of a wave (exact), and the library orders the units of the f16 pack so that silent ones share chunks")
    else:
        out["frac_issued"] = None
    out.update({"traffic": traffic,
                "traffic_source": (f"profiles/pmc_traffic.json = profiles/{prof_round}_pmc_hbm*.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, "
                                   f"FETCH doubled per the gfx950 note; collected in round {prof_round} by tools/profile_refresh.sh; not re-collected by this run)") if traffic else None,
                "launches": int(rr["p1_launches"]), "launch_sampling": f"HIP events on the context's stream around every {args.prof_stride}-th launch inside the timed blocks",
                "avg_launch_ms": avg_ms, "flops_per_launch": flops_launch,
                "profile": "profiles/%s_kernel_trace_stats%s.txt (rocprofv3 --kernel-trace --stats of this command with --path %s)" %
                           (ROUND, "_fp32" if fp32 else "", "fp32" if fp32 else "screened")})
    return out


SCREEN_KEYS = ("active", "eps", "candidates_per_rollout_step", "fallbacks", "audit_max_err", "audit_rows_per_rollout_step", "calibrations",
               "suspended", "sweeps", "sweep_max_err")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="franka_shelf_1024x32", choices=sorted(WORKLOADS))
    ap.add_argument("--kernels", type=int, default=10, help="active RBF navigation kernels K")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank owns the workload's N rollouts; strong: the N rollouts are split over the ranks")
    ap.add_argument("--path", choices=("fp32", "screened"), default=None,
                    help="the PRIMARY measurement (`value`): fp32 = the all-fp32 step, the reference's arithmetic throughout (default); "
                         "screened = the opt-in step (f16 screening of the first-pass rows + fp32 re-evaluation, omds_set_screening(ctx, 2, 0)).  The other one "
                         "is reported beside it unless --no-secondary")
    ap.add_argument("--screening", type=int, default=None, choices=(-1, 0, 1, 2),
                    help="older spelling of --path: 0 = fp32, 2 (or -1) = screened where it pays, 1 = screening forced on")
    ap.add_argument("--dense-pass1", action="store_true", help="OMDS_FLAG_DENSE_PASS1: k_pass1 without its exact zero-skip (A/B runs; same bits)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other path and the workloads / sweeps reported under 'also'")
    ap.add_argument("--allow-host-collectives", action="store_true",
                    help="when the RCCL communicator cannot be formed, exchange the update's sums through the launcher's gloo group instead of "
                         "exiting non-zero; the line is labelled config.collectives = gloo-host (...)")
    ap.add_argument("--try-rccl", action="store_true", help="with --share-gpu: attempt the RCCL communicator anyway (test of the failure path with OMDS_RCCL_LIB)")
    ap.add_argument("--reps", type=int, default=10, help="timed blocks of --steps iterations each; value = median block")
    ap.add_argument("--prof-stride", type=int, default=8,
                    help="HIP events around every n-th launch of the dominant kernel inside the timed region (an event record between two "
                         "launches idles the GPU for a few us: every launch costs 2.4 %% of value at 190 us per step, every 8th 0.2 %%)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test mode for a 1-GPU box: all ranks use GPU 0 and exchange through the host (gloo); never a scaling number")
    ap.add_argument("--time-fetch", action="store_true", help="(kept for older command lines: the fetch of all rollout tensors is always timed now)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.path is None:
        args.path = "fp32" if args.screening in (None, 0) else "screened"
    prim_scr = 0 if args.path == "fp32" else (args.screening if args.screening in (1, 2) else 2)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on this driver
    under_launcher = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if args.gpus > 1 and not under_launcher:
        self_launch(args, sys.argv[1:])      # never returns
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to file a {world}-rank run as {args.gpus} GPUs")
    from optimalmodulationds_amd import _lib
    ndev = _lib.device_count()      # the C-ABI's own count: no PyTorch on the device
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback): omds_device_count() = 0")
    if args.share_gpu:
        local_rank = 0
    elif local_rank >= ndev:
        raise SystemExit(f"bench.py: rank {rank} (local rank {local_rank}) has no GPU of its own: this box has {ndev}; --gpus {args.gpus} needs one GPU per rank")
    w0 = WORKLOADS[args.workload]
    if args.scaling == "strong" and w0["N"] % world:
        raise SystemExit(f"--scaling strong: {w0['N']} rollouts do not split over {world} ranks")
    n_local = w0["N"] // world if args.scaling == "strong" else w0["N"]
    n_total = n_local * world
    L = Launch(rank, world, local_rank)
    if under_launcher:
        # launcher plumbing only (RCCL id, barriers, max over ranks of the elapsed time, the dynamic workload's kernel broadcast); the
        # data path's collectives are the library's own RCCL calls (csrc/comm.hip).  torch is imported for gloo -- CPU tensors only
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        L.dist, L.torch = dist, torch

    shard = n_local if args.scaling == "strong" else None
    r = measure(args, L, args.workload, args.steps, args.warmup, time_fetch=True, reps=max(1, args.reps), screening=prim_scr, rollouts=shard)
    w, W, b, obs, q0, qf, H, K = (r[k] for k in ("w", "W", "b", "obs", "q0", "qf", "H", "K"))
    prim_fp32 = args.path == "fp32"

    other = None
    also = None
    if not args.no_secondary:
        # the same iterations on the other path (fp32 primary: the opt-in screened step; screened primary: screening off)
        ro = measure(args, L, args.workload, args.steps, 1, reps=3 if prim_fp32 is False else max(3, args.reps // 2), screening=(2 if prim_fp32 else 0), rollouts=shard)
        other = dict(rate(ro, n_total), roofline=roofline(args, ro, args.workload, not prim_fp32), screening=ro["scr"])
        also = []

        def both(wl2, st2, rp2, **kw):
            """One `also` entry: the workload on the all-fp32 step (`value`, the library's default) and on the opt-in screened step (`value_screened`;
            absent where the library does not screen: few obstacles, N x O < 65536)."""
            nt = (kw.get("rollouts") or WORKLOADS[wl2]["N"]) * world
            r32 = measure(args, L, wl2, st2, 1, prof=False, reps=rp2, screening=0, **kw)
            e2 = dict({"workload": wl2, "unit": "rollout-steps/s", "steps": st2, "dtype": "f32", "active_kernels": r32["K"], "rollouts_per_gpu": r32["N"]}, **rate(r32, nt))
            rs = measure(args, L, wl2, st2, 1, prof=False, reps=rp2, screening=2, **kw)
            if rs["scr"]["active"]:
                rt = rate(rs, nt)
                e2.update({"value_screened": rt["value"], "ms_per_step_screened": rt["ms_per_step"], "rep_ms_per_step_screened": rt["rep_ms_per_step"],
                           "screening": {k2: rs["scr"][k2] for k2 in SCREEN_KEYS}})
            else:
                e2["value_screened"] = None      # omds_set_screening(ctx, 2, 0) keeps the fp32 step at this shape
            if WORKLOADS[wl2].get("dynamic"):
                e2["kernels_at_block_end"] = rs["k_trace"]   # K grows from 0 inside every timed block (one add_kernel per iteration at most)
            return e2

        # the other workloads and the sweeps are single-GPU context for the headline; a multi-rank run measures the headline on both paths
        # and nothing else (every extra context would form another RCCL communicator across all ranks)
        for wl2, st2, rp2 in (("planar7_1024x32", 10, 5), ("franka_shelf_4096x32", 5, 3), ("franka_dynamic_1024x32", 20, 3),
                              ("franka_tanh_4096x32", 5, 3), ("franka_shelf_4096x64", 3, 3), ("franka_shelf_8192x32", 3, 3)):
            if wl2 != args.workload and world == 1:
                also.append(both(wl2, st2, rp2))
        if world == 1:
            # SURVEY 8d "policy state for timing": K in {0, 10, 50} (policy.py:17 N_KERNEL_MAX = 50); the primary line is K = --kernels
            for k2 in (0, 50):
                if k2 != K:
                    also.append(dict(both(args.workload, 10, 3, kernels=k2), sweep="active_kernels"))
            # one GPU's share of the workload under strong scaling at 2 / 4 / 8 ranks (DESIGN.md section 6 builds its prediction on these)
            for g in (2, 4, 8):
                if w0["N"] % g == 0 and w0["N"] // g >= 64:
                    also.append(dict(both(args.workload, 10, 3, rollouts=w0["N"] // g), sweep="strong_scaling_shard", ranks=g))
    if rank == 0:
        act = w.get("act", "relu")
        rt = rate(r, n_total)
        scr_sentence = ("fp32 outputs (every distance, gradient, velocity and cost the step returns comes from fp32 kernels); the N x O first-pass "
                        "evaluations that only feed the obstacle selection are screened in f16 and the candidates re-evaluated in fp32 -- identity with "
                        f"the all-fp32 step is conditional on a measured bound eps (profiles/r04_screen_error_hist.txt: 0 of > 1e11 unevaluated pairs above eps / 2)")
        out = {
            "metric": "modulated rollout-steps/sec", **{k2: v2 for k2, v2 in rt.items() if k2 not in ("ms_per_step", "rep_ms_per_step")},
            "unit": "rollout-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rt["ms_per_step"], "rep_ms_per_step": rt["rep_ms_per_step"], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            # dtype = the arithmetic of the measured step.  fp32 path: every network evaluation in fp32.  Screened path: fp32 outputs, f16
            # arithmetic in the N x O first-pass rows that only pick the candidates
            "dtype": "f32" if (prim_fp32 or not r["scr"]["active"]) else "f32 out / f16 screen", "data": "synthetic",
            "config": {"workload": args.workload,
                       "path": ("all-fp32 step (omds_set_screening(0)): every network evaluation in the reference's arithmetic; value_screened / "
                                "roofline.screened = the opt-in screened step (omds_set_screening(ctx, 2, 0)) on the same iterations") if prim_fp32 else
                               ("the opt-in screened step; value_fp32_only / roofline.fp32_only = the same iterations with every row in fp32 (the library's default)"),
                       "precision": "fp32 throughout" if (prim_fp32 or not r["scr"]["active"]) else scr_sentence,
                       "rollouts_per_gpu": r["N"], "rollouts_total": n_total, "horizon": H, "obstacles": int(obs.shape[0]),
                       "n_closest": w["k"], "active_kernels": K, "network": "x".join(str(x.shape[1]) for x in W) + f"x{W[-1].shape[0]} {act} "
                       + ("(shipped reference weights)" if act == "relu" else "(seeded synthetic weights)"),
                       "parallelism": f"rollout-sharded x{world}", "collectives": L.collectives,
                       # the world size the library's own RCCL communicator reports (omds_comm_info on every rank, required to equal --gpus;
                       # 1 = no communicator: one rank) and the HIP devices visible to each rank's process (omds_device_count)
                       "rccl_ranks": (L.rccl or {}).get("rccl_ranks", 1 if world == 1 else 0),
                       "devices_seen": (L.rccl or {}).get("devices_seen", [int(ndev)]),
                       "device_of_each_process": (L.rccl or {}).get("device_of_each_process", [int(L.local_rank)]),
                       # the device is reached through the C-ABI alone; torch (when imported at all: gloo plumbing, CPU baseline) stays on the CPU
                       "pytorch_on_device": bool("torch" in sys.modules and sys.modules["torch"].cuda.is_initialized())},
            "roofline": roofline(args, r, args.workload, prim_fp32),
        }
        if other is not None:
            ro_f = other["roofline"]
            brief = {"kernel": ro_f["kernel"], "pipe": ro_f["pipe"], "frac": ro_f["frac"], "frac_issued": ro_f["frac_issued"], "achieved": ro_f["achieved"],
                     "peak": ro_f["peak"], "avg_launch_ms": ro_f["avg_launch_ms"], "launches": ro_f["launches"], "traffic": ro_f["traffic"],
                     "value": other["value"], "ms_per_step": other["ms_per_step"], "profile": ro_f["profile"]}
            if prim_fp32:
                out["value_screened"] = other["value"]
                out["screened"] = dict(other, dtype="f32 out / f16 screen", precision=scr_sentence)
                out["roofline"]["screened"] = dict(brief, dtype="f32 out / f16 screen")
            else:
                out["value_fp32_only"] = other["value"]
                out["fp32_only"] = other
                out["roofline"]["fp32_only"] = brief
        out["screening"] = r["scr"]   # of the primary run (inactive on the fp32 path)
        out["fetch_all_rollouts_ms"] = r["fetch_ms"]
        if also is not None:
            out["also"] = also
        if not args.no_cpu_baseline and world == 1 and act == "relu":   # oracle/torch_baseline.py restates the shipped (ReLU) path
            out["cpu_baseline"] = cpu_baseline(w, W, b, obs, q0, qf, r["dh"], r["qmin"], r["qmax"], K, H)
        print(json.dumps(out))
    if L.use_dist:
        L.dist.destroy_process_group()


if __name__ == "__main__":
    main()

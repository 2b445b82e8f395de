"""GPU tests of the rollout-sharded update: two device contexts that each hold half of the rollouts must
reproduce one context holding all of them; the native RCCL path (csrc/comm.hip) runs with a single-rank communicator
(a 1-GPU box cannot host two RCCL ranks); and BASELINE configs[3]'s per-GPU shard (4096 rollouts x 64 horizon) runs
and is re-derived by the oracle at sampled states."""
import os

import numpy as np
import pytest

from helpers import assert_velocity_plain, weights_path
from oracle import omds_oracle as orc

pytestmark = pytest.mark.gpu


def _setup(N, H, k=5, K=6, seed=3, scene="shelf"):
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path("franka"))
    obs = scenes.shelf_scene()
    q0, qf, dh = scenes.FRANKA_Q0, scenes.FRANKA_QF, scenes.franka_dh_params()
    qmin, qmax = np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32)

    def make(n_traj):
        e = Engine(7, n_traj, H, k, max_obs=512)
        e.set_screening(2)      # the opt-in screened step where it pays (the library's default is the all-fp32 step)
        e.set_mlp(m.W, m.b)
        e.set_obstacles(obs)
        e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.5, 0.01, 0b111
        e.push_params()
        e.set_ds(qf)
        e.set_cost(dh, qmin, qmax)
        return e

    rng = np.random.RandomState(seed)
    s = (np.arange(K) + 0.5) / max(K, 1)
    mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
    sg_c = np.ones(K, np.float32)
    al_c = rng.standard_normal((K, 7)).astype(np.float32)
    return m, obs, q0, qf, dh, qmin, qmax, make, mu_c, sg_c, al_c


@pytest.mark.parametrize("N,H,K", [(512, 8, 6), (96, 5, 0), (2048, 4, 50)])
def test_two_contexts_sum_to_one(N, H, K):
    """Shards [0, N/2) and [N/2, N) on two contexts of device 0, with the samples of the single N-rollout context split
    between them (sizes chosen so that the full and the half shape take the same path -- both screened or both not, same tile heights -- hence bit-identical rollouts): their cost sums and packed partial sums added on the host (what the all-reduce does) must reproduce
    the single context's update -- identical mask, means to 1e-6 -- and the MINLOC over the shards its best rollout."""
    from optimalmodulationds_amd.engine import apply_update, red_layout
    m, obs, q0, qf, dh, qmin, qmax, make, mu_c, sg_c, al_c = _setup(N, H, K=K)
    one = make(N)
    one.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=77, rollout_offset=0)
    mu, sg, al = one.get_policy_samples()
    one.propagate(q0)
    c_one = one.cost()
    ref_mu, ref_sg, ref_al, ref_mask, ref_w = one.weighted_update(0.1, 0.1, mu_c, sg_c, al_c, want_weights=True)
    qd_w, qd_b = one.get_qdot("weighted"), one.get_qdot("best")

    h = N // 2
    shards = []
    for r in range(2):
        e = make(h)
        if K:
            # device-side sampling with the shard's rollout_offset must give exactly the single context's samples
            e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=77, rollout_offset=r * h)
            smu, ssg, sal = e.get_policy_samples()
            assert np.array_equal(smu, mu[r * h:(r + 1) * h]) and np.array_equal(sal, al[r * h:(r + 1) * h])
            assert np.array_equal(ssg, sg[r * h:(r + 1) * h])
        else:
            e.sample_policy(mu_c, sg_c, al_c, 0, 0, 0, 0, seed=77, rollout_offset=r * h)
        e.propagate(q0)
        c = e.cost()
        assert np.array_equal(c, c_one[r * h:(r + 1) * h])       # rollouts do not depend on which context holds them
        shards.append(e)
    cs = sum(e.cost_sum() for e in shards)                        # all-reduce SUM #1
    assert cs[1] == N
    reds = [e.local_sums(cs[0], cs[1], include_rollout0=(r == 0)) for r, e in enumerate(shards)]
    lay = red_layout(K, 7)
    red = reds[0].copy()
    red[:lay["n_sum"]] = reds[0][:lay["n_sum"]] + reds[1][:lay["n_sum"]]   # all-reduce SUM #2
    nmu, nsg, nal, mask = apply_update(K, 7, H, red, float(cs[1]), 0.1, 0.1, mu_c, sg_c, al_c)
    assert np.array_equal(mask, ref_mask)
    if K:
        for a, b, what in ((nmu, ref_mu, "mu"), (nsg, ref_sg, "sigma"), (nal, ref_al, "alpha")):
            assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max()), what
    qw = red[lay["qdot"]:lay["qdot"] + 7] / red[0]
    assert np.abs(qw - qd_w).max() <= 2e-6
    best = np.stack([r_[lay["n_sum"]:] for r_ in reds])           # all-gather + MINLOC
    assert np.array_equal(best[int(np.argmin(best[:, 0])), 1:], qd_b)
    for e in shards + [one]:
        e.close()


def test_eight_shards_of_1024_reproduce_one_context_of_8192x32():
    """BASELINE configs[4]'s shape -- 8192 rollouts x 32 steps over 8 GPUs -- emulated on one: eight contexts of 1024 rollouts (the
    per-GPU shard, rollout_offset = rank * 1024, the moving shelf re-sent before the propagate like the dynamic workload does) and
    ONE context holding all 8192.  Every shard's rollouts are the corresponding rows of the big context bit for bit (both shapes
    take the screened step); the host-mediated sums over the eight (what the two all-reduces and the MINLOC gather do) reproduce
    the single context's update: identical mask, means and weighted velocity to 1e-6, the same best rollout."""
    from optimalmodulationds_amd.engine import apply_update, red_layout
    N, H, K, G = 8192, 32, 10, 8
    m, obs, q0, qf, dh, qmin, qmax, make, mu_c, sg_c, al_c = _setup(N, H, K=K)
    moved = obs.copy()
    moved[:, 1] += 0.05 * np.sin(0.3 * 7)
    one = make(N)
    one.set_obstacles(moved)
    one.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=4242, rollout_offset=0)
    one.propagate(q0)
    r_one = one.get_rollouts(want=("all_traj", "closest_dist_all", "qdot"))
    c_one = one.cost()
    ref_mu, ref_sg, ref_al, ref_mask, ref_qw, ref_qb, nt = one.weighted_update_sharded(0.1, 0.1, mu_c, sg_c, al_c, want_best=True)
    assert nt == N and one.screen_stats()["active"]
    h = N // G
    cs, shards = np.zeros(2, np.float64), []
    for r in range(G):
        e = make(h)
        e.set_obstacles(moved)
        e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=4242, rollout_offset=r * h)
        e.propagate(q0)
        rr = e.get_rollouts(want=("all_traj", "closest_dist_all", "qdot"))
        for key in rr:
            assert np.array_equal(rr[key], r_one[key][r * h:(r + 1) * h]), (r, key)
        assert np.array_equal(e.cost(), c_one[r * h:(r + 1) * h])
        st = e.screen_stats()      # a shard's first propagate may trip the guard of its fresh calibration once (redone in fp32: the same bits)
        print(r, {k2: st[k2] for k2 in ("eps", "max_err_seen", "audit_max_err", "fallbacks_by_error", "fallbacks_by_slack", "fallbacks_by_overflow")})
        assert st["active"] and not st["suspended"] and st["fallbacks"] <= 1, st
        cs += e.cost_sum()                                            # all-reduce SUM #1 over the 8 shards
        shards.append(e)
    assert cs[1] == N
    reds = [e.local_sums(np.float32(cs[0]), np.float32(cs[1]), include_rollout0=(r == 0)) for r, e in enumerate(shards)]
    lay = red_layout(K, 7)
    red = reds[0].copy()
    red[:lay["n_sum"]] = np.sum([x[:lay["n_sum"]] for x in reds], axis=0, dtype=np.float32)   # all-reduce SUM #2
    nmu, nsg, nal, mask = apply_update(K, 7, H, red, float(cs[1]), 0.1, 0.1, mu_c, sg_c, al_c)
    assert np.array_equal(mask, ref_mask)
    for a, b, what in ((nmu, ref_mu, "mu"), (nsg, ref_sg, "sigma"), (nal, ref_al, "alpha")):
        assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max()), what
    assert np.abs(red[lay["qdot"]:lay["qdot"] + 7] / red[0] - ref_qw).max() <= 2e-6
    best = np.stack([x[lay["n_sum"]:] for x in reds])                 # all-gather + MINLOC (lowest rank on ties)
    assert np.array_equal(best[int(np.argmin(best[:, 0])), 1:], ref_qb)
    for e in shards + [one]:
        e.close()


def test_native_rccl_single_rank_matches_local_update():
    """omds_comm_init_rank (world 1) + omds_weighted_update_sharded: the RCCL all-reduces / all-gather run on the
    context stream on device buffers; with one rank they must leave every number of the local update unchanged."""
    from optimalmodulationds_amd.engine import Engine
    N, H, K = 256, 6, 8
    m, obs, q0, qf, dh, qmin, qmax, make, mu_c, sg_c, al_c = _setup(N, H, K=K)
    e = make(N)
    e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=5)
    e.propagate(q0)
    e.cost(fetch=False)
    ref = e.weighted_update(0.1, 0.1, mu_c, sg_c, al_c)
    qd_w, qd_b = e.get_qdot("weighted"), e.get_qdot("best")
    assert e.comm_info() == (0, 1)
    e.comm_init(Engine.comm_unique_id(), 0, 1)
    assert e.comm_info() == (0, 1)
    for _ in range(3):   # repeated collectives on the same communicator
        mu, sg, al, mask, qw, qb, nt = e.weighted_update_sharded(0.1, 0.1, mu_c, sg_c, al_c, want_best=True)
        assert nt == N
        assert np.array_equal(mask, ref[3])
        assert np.array_equal(mu, ref[0]) and np.array_equal(sg, ref[1]) and np.array_equal(al, ref[2])
        assert np.array_equal(qw, qd_w) and np.array_equal(qb, qd_b)
    e.comm_destroy()
    assert e.comm_info() == (0, 1)
    mu, sg, al, mask, qw, qb, nt = e.weighted_update_sharded(0.1, 0.1, mu_c, sg_c, al_c, want_best=True)   # no communicator: local
    assert np.array_equal(mu, ref[0]) and np.array_equal(qb, qd_b)
    e.close()


def test_native_rccl_when_the_library_is_loaded_before_torch():
    """Load order: a process that loads libomds_hip.so (bound to the ROCm install's HIP runtime) BEFORE importing torch
    must still get a working communicator -- the library picks the RCCL beside the HIP runtime it is bound to, not the copy
    a later ``import torch`` maps (that one opens a second HSA runtime and fails with 'no ROCm-capable device')."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from optimalmodulationds_amd.engine import Engine\n"
        "assert 'torch' not in sys.modules\n"
        "e = Engine(7, 64, 2, 1, max_obs=8)\n"
        "import torch\n"
        "e.comm_init(Engine.comm_unique_id(), 0, 1)\n"
        "assert e.comm_info() == (0, 1)\n"
        "e.comm_destroy(); e.close(); print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_shard_4096x64_config4():
    """BASELINE configs[3] per GPU: 4096 rollouts x 64 horizon on the shelf scene (one of 8 shards; rollout_offset of
    rank 3).  Size-independent checks: finite outputs; H network evaluations; sampled (rollout, step) states re-derived
    by the oracle; cost / update recomputed by the oracle from the device's rollouts; the sharded-update entry point."""
    N, H, K, k = 4096, 64, 10, 5
    m, obs, q0, qf, dh, qmin, qmax, make, mu_c, sg_c, al_c = _setup(N, H, K=K)
    e = make(N)
    e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=11, rollout_offset=3 * N)
    mu, sg, al = e.get_policy_samples()
    assert not np.array_equal(al[0], al_c)       # not the owner of global rollout 0: its rollout 0 is noised
    e.propagate(q0)
    r = e.get_rollouts()
    for key, v in r.items():
        assert np.isfinite(v).all(), key
    assert r["all_traj"].shape == (N, H, 7) and np.array_equal(r["all_traj"][:, 0], np.broadcast_to(q0, (N, 7)))
    rng = np.random.RandomState(0)
    S = 192
    tt, hh = rng.randint(0, N, S), rng.randint(0, H, S)
    hh[:16] = H - 1                               # the last evaluation is computed but not integrated (MPPI.py:220)
    q = r["all_traj"][tt, hh]
    d, g, _, idx = orc.distance_repulsion_nn(m, q, obs, k, [0, 1, 2])
    st = orc.modulation_step(q, qf, d, g, mu[tt], sg[tt], al[tt], orc.Params(dst_thr=0.01))
    assert np.abs(r["closest_dist_all"][tt, hh] - (d - np.float32(0.01))).max() <= 1e-6
    assert np.abs(r["kernel_val_all"][tt, hh] - st["phi"]).max() <= 1e-5
    nxt = hh + 1 < H             # every sampled row with a next state
    vel = (r["all_traj"][tt[nxt], hh[nxt] + 1] - q[nxt]) / np.float32(0.5)
    assert_velocity_plain(vel, q[nxt], qf, d[nxt], g[nxt], mu[tt][nxt], sg[tt][nxt], al[tt][nxt], orc.Params(dst_thr=0.01),
                          "shard 4096 x 64, sampled rows", pad=4e-6 * max(1.0, float(np.abs(q).max())) / 0.5, family="franka")
    cost = e.cost()
    ocost, _ = orc.evaluate_costs(r["all_traj"], r["closest_dist_all"], qf, dh, qmin, qmax)
    assert np.abs(cost - ocost).max() <= 1e-5 * max(1.0, np.abs(ocost).max())
    nmu, nsg, nal, mask, qw, _, nt = e.weighted_update_sharded(0.1, 0.1, mu_c, sg_c, al_c)
    omu, osg, oal, omask, _ = orc.shift_policy_means(cost, r["kernel_val_all"], r["kernel_activations"], mu_c, sg_c, al_c,
                                                     mu, sg, al, 0.1, 0.1)
    assert nt == N and np.array_equal(mask, omask)
    assert np.abs(nal - oal).max() <= 2e-5 * max(1.0, np.abs(oal).max())
    e.close()


def test_communicator_and_screening_state_survive_obstacle_growth():
    """A rank whose scene outgrows its obstacle buffers (MPPI.update_obstacles takes any count, MPPI.py:347-350) must stay in
    the collective: omds_set_obstacles grows the buffers inside the context -- the handle, the RCCL communicator, the
    screening mode and the policy samples survive -- and the sharded update afterwards is the update of the new scene."""
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
    from optimalmodulationds_amd.engine import Engine
    N, H, K = 512, 6, 8
    m, obs, q0, qf, dh, qmin, qmax, make, mu_c, sg_c, al_c = _setup(N, H, K=K)
    e = Engine(7, N, H, 5, max_obs=64)
    e.set_mlp(m.W, m.b)
    e.set_obstacles(obs[::8][:36])
    e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.5, 0.01, 0b111
    e.push_params()
    e.set_ds(qf)
    e.set_cost(dh, np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32))
    e.set_screening(1)
    e.comm_init(Engine.comm_unique_id(), 0, 1)
    e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=5)
    e.propagate(q0)
    assert e.comm_active() and e.screen_stats()["active"] and e.max_obs == 64
    e.set_obstacles(obs)                                   # 294 > 64: grows
    assert e.max_obs >= 294 and e.comm_active()
    st = e.screen_stats()
    assert st["active"], st                                # the screening request survived (the bound is recalibrated: new scene)
    e.propagate(q0)
    e.cost(fetch=False)
    got = e.weighted_update_sharded(0.1, 0.1, mu_c, sg_c, al_c, want_best=True)
    assert e.screen_stats()["calibrations"] == 2
    ref_e = make(N)                                        # a fresh context on the big scene, fp32 step, no communicator
    ref_e.set_screening(0)
    ref_e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=5)
    ref_e.propagate(q0)
    ref_e.cost(fetch=False)
    ref = ref_e.weighted_update_sharded(0.1, 0.1, mu_c, sg_c, al_c, want_best=True)
    for a, b in zip(got[:6], ref[:6]):
        assert np.array_equal(a, b)
    e.comm_destroy()
    e.close()
    ref_e.close()


def test_facade_shift_policy_means_uses_the_native_sharded_update():
    """MPPI.init_comm + MPPI.shift_policy_means / get_qdot (the reference's API, MPPI.py:319-345): with a (single-rank)
    communicator the facade goes through omds_weighted_update_sharded and gives the numbers of the local update."""
    import torch
    from optimalmodulationds_amd import MPPI, LinDS, RobotSdfCollisionNet, scenes
    nn = RobotSdfCollisionNet(10, 9, [], [256] * 4)
    nn.load_weights(weights_path("franka"), {'device': 'cpu', 'dtype': torch.float32})
    outs = []
    for with_comm in (False, True):
        ds = LinDS(torch.tensor(scenes.FRANKA_QF))
        mppi = MPPI(torch.tensor(scenes.FRANKA_Q0), torch.tensor(scenes.FRANKA_QF), torch.tensor(scenes.franka_dh_params()),
                    torch.tensor(scenes.shelf_scene()[::6]), 0.5, 6, 256, [ds], None, nn, 5, seed=7)
        mppi.dst_thr, mppi.ker_thr = 0.01, 0.1
        mppi.Policy.alpha_s = 3.0
        rng = np.random.RandomState(0)
        for i in range(4):
            mppi.Policy.add_kernel(torch.tensor((scenes.FRANKA_Q0 + 0.2 * rng.standard_normal(7)).astype(np.float32)), 1.0, torch.eye(7))
        if with_comm:
            assert mppi.init_comm() == (0, 1) and mppi._engine.comm_active()
        mppi.Policy.sample_policy()
        mppi.propagate()
        mppi.get_cost()
        mppi.shift_policy_means()
        outs.append((mppi.Policy.mu_c.clone(), mppi.Policy.alpha_c.clone(), mppi.update_mask.clone(), mppi.get_qdot('best').clone(),
                     mppi.get_qdot('weighted').clone(), mppi.qdot_weighted.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("workload", ["franka_shelf_1024x32", "franka_dynamic_1024x32"])
def test_bench_under_torchrun_single_rank_uses_rccl(workload):
    """The exact command line the driver uses for the scaling bench, at one rank: a child process started from here (the
    launcher and the worker initialise the GPU themselves), rc 0, one JSON line, collectives = rccl.  The dynamic workload also
    runs its per-iteration kernel adding through the launcher's broadcast (rank 0 decides, every rank installs the same kernel)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533" if workload.startswith("franka_shelf") else "29534", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--reps", "2",
           "--no-cpu-baseline", "--no-secondary", "--workload", workload]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["config"]["collectives"] == "rccl" and d["value"] > 1e5, d
    assert d["scaling"] == "weak" and d["steps"] == 3 and d["reps"] == 2
    assert d["config"]["pytorch_on_device"] is False      # gloo plumbing only: PyTorch never initialised the GPU in the measured process
    # the line certifies its own communicator: the world size the LIBRARY's RCCL communicator reported (omds_comm_info on every rank,
    # required to equal --gpus) and the devices each rank's process saw through the C-ABI
    assert d["config"]["rccl_ranks"] == 1 and d["config"]["device_of_each_process"] == [0], d["config"]
    assert len(d["config"]["devices_seen"]) == 1 and d["config"]["devices_seen"][0] >= 1, d["config"]


@pytest.mark.parametrize("mode", ["share", "rccl_missing", "rccl_missing_strict"])
def test_bench_two_ranks_on_one_gpu(mode):
    """World size 2 on a 1-GPU box THROUGH THE SELF-LAUNCHER: `python bench.py --gpus 2 --share-gpu` starts its own two ranks (a child
    torch.distributed.run, before the parent touches the GPU) and relays rank 0's line.  Both ranks on GPU 0, sums through the
    launcher's gloo group (two ranks on one GPU cannot form a RCCL communicator) -- the barrier, the max over ranks, the rank-0 kernel
    decision + broadcast of the dynamic workload and the whole-job value all run.  'rccl_missing': the ranks attempt the RCCL
    communicator, cannot load the library (OMDS_RCCL_LIB points nowhere), and -- because --allow-host-collectives asks for it -- agree
    on the host-mediated path and say so in the line.  'rccl_missing_strict': the same without that flag: no number, exit non-zero."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    extra = []
    if mode != "share":
        env["OMDS_RCCL_LIB"] = "/nonexistent/librccl.so"
        extra = ["--try-rccl"] + (["--allow-host-collectives"] if mode == "rccl_missing" else [])
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--reps", "2", "--no-cpu-baseline",
           "--no-secondary", "--share-gpu", "--path", "screened", "--workload", "franka_dynamic_1024x32"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    if mode == "rccl_missing_strict":
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")], r.stdout[-2000:]
        assert "could not be formed" in r.stderr and "refusing" in r.stderr and "no number is reported" in r.stderr, r.stderr[-3000:]
        return
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the line, the launcher relays it once"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 1e5, d
    assert d["config"]["parallelism"] == "rollout-sharded x2" and d["config"]["rollouts_total"] == 2048
    coll = d["config"]["collectives"]
    assert d["config"]["rccl_ranks"] == 0, d["config"]      # two ranks, no RCCL communicator: the line says so in this field too
    if mode == "share":
        assert coll.startswith("gloo-host (--share-gpu"), coll
    else:
        assert coll.startswith("gloo-host (RCCL unavailable: rank 0:"), coll
        assert "RCCL communicator unavailable" in r.stderr, r.stderr[-2000:]
    # whole-job value = both ranks' rollouts over the slowest rank's time (medians of two blocks: of the rates / of the times -- the mean of
    # two rates is not the rate of the mean time, hence the 2 % when the blocks differ by 10 %)
    assert abs(d["value"] - 2 * 1024 * 32 / (d["ms_per_step"] * 1e-3)) < 2e-2 * d["value"]
    assert d["value_min"] <= 2 * 1024 * 32 / (max(d["rep_ms_per_step"]) * 1e-3) * (1 + 1e-4)


def test_bench_strong_scaling_splits_the_rollouts_over_the_ranks():
    """--scaling strong at two ranks (sharing the one GPU of the box): the workload's 1024 rollouts are split 512 + 512, the line counts
    1024 rollouts per iteration, and rank 1's samples start at rollout offset 512."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--reps", "2", "--no-cpu-baseline",
           "--no-secondary", "--share-gpu", "--scaling", "strong"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["scaling"] == "strong" and d["n_gpus"] == 2 and d["dtype"] == "f32", d
    assert d["config"]["rollouts_per_gpu"] == 512 and d["config"]["rollouts_total"] == 1024
    assert abs(d["value"] - 1024 * 32 / (d["ms_per_step"] * 1e-3)) < 2e-2 * d["value"]

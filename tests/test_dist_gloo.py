"""The N>1 path on CPU: two processes over gloo run the real ``dist.sharded_update`` (cost-sum
all-reduce -> packed partial sums all-reduce + best all-gather -> omds_apply_update) on two
rollout shards whose partial sums come from the oracle's arithmetic, and must reproduce the
single-process update captured from the reference."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from helpers import ROOT, assert_close, load


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, outdir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from test_capi_cpu import _packed_from_oracle
    from optimalmodulationds_amd.dist import sharded_update
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = load(name)
    K, n, H, N = int(fx["K"]), fx["q0"].shape[0], int(fx["H"]), int(fx["N"])
    lo, hi = rank * N // world, (rank + 1) * N // world
    pre = "it0_"
    cost = fx[pre + "cost"]

    def cost_sum():
        return np.array([cost[lo:hi].sum(dtype=np.float32), hi - lo], np.float32)

    def local_sums(sum_cost, n_total, include0):
        assert abs(sum_cost - float(cost.sum(dtype=np.float32))) < 1e-3 * abs(sum_cost) and n_total == N
        assert include0 == (rank == 0)
        return _packed_from_oracle(fx, pre, lo, hi, include0)

    mu, sg, al, mask, qw, qb = sharded_update(cost_sum, local_sums, K, n, H, float(fx["policy_upd_rate"]),
                                              float(fx["ker_thr"]), fx[pre + "mu_c"], fx[pre + "sigma_c"],
                                              fx[pre + "alpha_c"])
    np.savez(os.path.join(outdir, f"r{rank}.npz"), mu=mu, sg=sg, al=al, mask=mask, qw=qw, qb=qb)
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["franka_shelf_K6", "planar7_K4", "franka_sub40_K50", "franka_shelf_K0"])
def test_two_rank_update_matches_single_process(tmp_path, name):
    import __graft_entry__ as g
    g.build()
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    fx = load(name)
    outs = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for k in ("mu", "sg", "al", "mask", "qw", "qb"):
        assert np.array_equal(outs[0][k], outs[1][k]), f"ranks disagree on {k}"
    o = outs[0]
    assert int(o["mask"].sum()) == int(fx["it0_n_updated"])
    assert_close(o["mu"], fx["it0_mu_c_new"], 1e-5, "mu_c")
    assert_close(o["sg"], fx["it0_sigma_c_new"], 1e-5, "sigma_c")
    assert_close(o["al"], fx["it0_alpha_c_new"], 1e-5, "alpha_c")
    assert_close(o["qw"], fx["it0_qdot_weighted"], 1e-5, "weighted qdot")
    assert_close(o["qb"], fx["it0_qdot_best"], 1e-6, "best qdot")


def _worker_no_rccl(rank, world, port, outdir, which):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    if which == "all" or rank == 1:
        os.environ["OMDS_RCCL_LIB"] = "/nonexistent/librccl-not-here.so"
    import torch.distributed as dist
    from optimalmodulationds_amd import _lib
    from optimalmodulationds_amd.dist import init_native_comm
    from optimalmodulationds_amd.engine import Engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Ctx:                       # the two calls init_native_comm makes; no GPU context behind them
        comm_unique_id = staticmethod(Engine.comm_unique_id)

        def comm_init(self, uid, rank, world):
            raise AssertionError("no rank may enter the collective init while one of them has no RCCL")

    msg = ""
    try:
        init_native_comm(Ctx())
    except _lib.OmdsError as e:
        msg = str(e)
    flags = [None] * world           # the agreement step bench.py runs behind the attempt
    dist.all_gather_object(flags, bool(msg))
    with open(os.path.join(outdir, f"r{rank}.txt"), "w") as f:
        f.write(f"{flags}|{msg}")
    dist.destroy_process_group()


@pytest.mark.parametrize("which", ["all", "rank1"])
def test_a_missing_rccl_fails_on_every_rank_instead_of_hanging_the_others(tmp_path, which):
    """librccl is not loadable on every rank / on rank 1 only: every rank probes the loader (omds_comm_probe) and the ranks agree
    BEFORE any of them enters the collective init, so all of them get the same error naming the rank(s) without RCCL -- nobody
    blocks inside ncclCommInitRank (or in the id broadcast) until the launcher's timeout."""
    import __graft_entry__ as g
    g.build()
    world = 2
    mp.spawn(_worker_no_rccl, args=(world, _free_port(), str(tmp_path), which), nprocs=world, join=True)
    for r in range(world):
        flags, msg = open(tmp_path / f"r{r}.txt").read().split("|", 1)
        assert flags == "[True, True]" and "RCCL not available" in msg and "librccl-not-here" in msg and "rank 1:" in msg, (r, flags, msg)
        assert ("rank 0:" in msg) == (which == "all"), msg

"""Multi-GPU sharding of the MPPI iteration (new work; the reference is single-process).

Rollouts are independent given (q_cur, obstacles, weights, policy means), so each rank (one
process per GPU) owns a contiguous block of N_local rollouts and nothing is exchanged inside
``propagate()`` or the cost.  The only exchange is the cost-weighted update
(MPPI.shift_policy_means / get_qdot, ds_mppi/functions/MPPI.py:319-345):

  1. all-reduce SUM of [sum cost, N_local]            (8 bytes)      -> global beta
  2. all-reduce SUM of the packed partial sums        (<= 3.4 KB)    -> update of mu/sigma/alpha
     + all-gather of (min cost, qdot of the arg-min)  (for get_qdot('best'))

Both are latency-bound.  On GPUs the exchange is NATIVE: the library owns a RCCL communicator per
context and runs the all-reduces on the context stream on device buffers (csrc/comm.hip,
``omds_weighted_update_sharded``); ``init_native_comm`` below only bootstraps it (the 128-byte id
travels over whatever process group the launcher has, e.g. gloo under torchrun).
``sharded_update`` is the host-mediated form of the same exchange (``omds_cost_sum`` /
``omds_local_sums`` + ``torch.distributed`` all-reduces), kept for launchers without RCCL and for
the world-size-2 gloo tests; it is never chosen silently.  The final arithmetic is
``omds_apply_update`` (host C, no GPU needed)."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .engine import apply_update, red_layout


def init_native_comm(engine, group=None):
    """Creates the RCCL communicator of ``engine`` over the ranks of ``group`` (any torch.distributed
    backend -- it only carries the id).  Raises OmdsError (OMDS_ERR_RCCL) ON EVERY RANK if RCCL cannot be used on any of
    them: there is no fallback, and no rank is left waiting inside ncclCommInitRank for one that could not load the
    library (every rank probes the loader first -- omds_comm_probe -- and the ranks agree before the collective init).
    Without an initialised process group: a single-rank communicator."""
    from . import _lib as L
    if dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        probes = [None] * world
        dist.all_gather_object(probes, L.comm_probe(), group=group)
        bad = [f"rank {r}: {m}" for r, m in enumerate(probes) if m]
        if bad:
            raise L.OmdsError("omds error 3: RCCL is not usable on every rank -- " + "; ".join(bad))
        box = [None]
        if rank == 0:
            try:
                box = [engine.comm_unique_id()]
            except Exception as e:          # ncclGetUniqueId failed: the other ranks are waiting in the broadcast -- tell them
                box = [e]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if isinstance(box[0], Exception):
            raise box[0]
        uid = box[0]
    else:
        rank, world, uid = 0, 1, engine.comm_unique_id()
    engine.comm_init(uid, rank, world)
    return rank, world


def _dev_for_backend(group=None):
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def all_reduce_sum(x: np.ndarray, group=None) -> np.ndarray:
    if not dist.is_initialized():
        return x
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(_dev_for_backend(group))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def all_gather(x: np.ndarray, group=None) -> np.ndarray:
    if not dist.is_initialized():
        return x[None]
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(_dev_for_backend(group))
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(outs, t, group=group)
    return torch.stack(outs).cpu().numpy()


def sharded_update(cost_sum_fn, local_sums_fn, K, n, H, rate, ker_thr, mu_c, sigma_c, alpha_c, group=None, want_best=True):
    """One cost-weighted update over all shards.

    cost_sum_fn() -> [sum cost, N_local] of this shard;
    local_sums_fn(sum_cost_global, n_total, include_rollout0) -> packed partial buffer (layout
    ``engine.red_layout``).  Returns (mu, sigma, alpha, mask, qdot_weighted, qdot_best); with
    ``want_best=False`` only two collectives run and qdot_best is this shard's own best rollout."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    cs = all_reduce_sum(np.asarray(cost_sum_fn(), dtype=np.float32), group)
    red = np.asarray(local_sums_fn(float(cs[0]), float(cs[1]), rank == 0), dtype=np.float32).copy()
    lay = red_layout(K, n)
    assert red.shape[0] == lay["size"], (red.shape, lay)
    red[:lay["n_sum"]] = all_reduce_sum(red[:lay["n_sum"]], group)
    # get_qdot('best') needs a MINLOC over the shards; the planner iteration itself (frankaPlanner.py:132-145)
    # never asks for it, so callers that do not need it skip this third collective
    best = all_gather(red[lay["n_sum"]:], group) if want_best else red[None, lay["n_sum"]:]    # [G, 1+n]
    b = int(np.argmin(best[:, 0]))                          # lowest rank on ties, like a global argmin
    mu, sg, al, mask = apply_update(K, n, H, red, float(cs[1]), rate, ker_thr, mu_c, sigma_c, alpha_c)
    qdot_w = red[lay["qdot"]:lay["qdot"] + n] / red[0]
    return mu, sg, al, mask, qdot_w.astype(np.float32), best[b, 1:].astype(np.float32)


def shift_policy_means_sharded(mppi, group=None):
    """``MPPI.shift_policy_means`` over the shards of ``group`` in the HOST-MEDIATED form (torch.distributed all-reduces of
    the library's partial sums): for launchers without RCCL and the world-size-2 gloo tests.  On GPUs use
    ``mppi.init_comm(group)`` once and then plain ``mppi.shift_policy_means()`` -- the native RCCL exchange."""
    if mppi.cur_cost is None:
        mppi._push()
        mppi._engine.cost(fetch=False)
        mppi.cur_cost = True
    P, e = mppi.Policy, mppi._engine
    K = P.n_kernels
    mu, sg, al, mask, qw, qb = sharded_update(e.cost_sum, e.local_sums, K, mppi.n_dof, mppi.dt_H, mppi.policy_upd_rate,
                                              mppi.ker_thr, P.mu_c.numpy(), P.sigma_c.numpy(), P.alpha_c.numpy(), group)
    if K > 0:
        P.mu_c[:K] = torch.from_numpy(mu)
        P.sigma_c[:K] = torch.from_numpy(sg)
        P.alpha_c[:K] = torch.from_numpy(al)
    mppi.update_mask = torch.from_numpy(mask)
    mppi.qdot_weighted, mppi.qdot_best = torch.from_numpy(qw), torch.from_numpy(qb)
    return 0, int(mask.sum())

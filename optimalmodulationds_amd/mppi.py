"""MPPI controller -- mirrors ``ds_mppi/functions/MPPI.py`` (class MPPI, lines 21-350).

Same constructor, methods and mutable attributes as the reference; every array computation of
the hot path (distance network forward/backward over rollouts x obstacles, modulation, policy,
Euler step, cost, cost-weighted reduction) runs in the HIP kernels behind ``Engine``.  Tensors
handed back to the caller are torch CPU tensors in the reference's layouts."""
from __future__ import annotations

import numpy as np
import torch

from .cost import Cost
from .engine import Engine
from .lazy import LazyRollout
from .policy import TensorPolicyMPPI


def _np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy().astype(np.float32)
    return np.asarray(x, dtype=np.float32)


def _qr_complete(g):
    """E = qr([g, e_2 .. e_n]).Q with column 0 := g, batched over the leading axes of g [..., n]."""
    n = g.shape[-1]
    A = np.broadcast_to(np.eye(n, dtype=np.float32), g.shape[:-1] + (n, n)).copy()
    A[..., 0] = g
    Q, _ = np.linalg.qr(A)
    Q = Q.astype(np.float32)
    Q[..., 0] = g
    return torch.from_numpy(Q)


class _LazyBasis:
    """Indexable stand-in for the reference's ``norm_basis`` tensor; ``[i, h]`` builds one basis,
    any other index (or ``.tensor()``) materialises the selection."""

    def __init__(self, normals):
        self._g = normals                                   # [N, H, n]: a LazyRollout (row fetches) or an array
        self.shape = tuple(normals.shape) + (normals.shape[-1],)

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        lead = tuple(int(i) if isinstance(i, torch.Tensor) and i.ndim == 0 else i for i in idx[:2])
        out = _qr_complete(np.asarray(self._g[lead], dtype=np.float32))
        return out[(Ellipsis,) + tuple(idx[2:])] if len(idx) > 2 else out

    def tensor(self):
        return _qr_complete(np.asarray(self._g, dtype=np.float32))

    def __torch_function__(self, func, types, args=(), kwargs=None):
        args = tuple(a.tensor() if isinstance(a, _LazyBasis) else a for a in args)
        return func(*args, **(kwargs or {}))

    def transpose(self, *a):
        return self.tensor().transpose(*a)

    def __matmul__(self, other):
        return self.tensor() @ (other.tensor() if isinstance(other, _LazyBasis) else other)


class MPPI:
    def __init__(self, q0, qf, dh_params, obs, dt, dt_H, N_traj, DS_ARRAY, dh_a, nn_model, n_closest_obs,
                 device=0, warmup=False, seed=1234, rollout_offset=0, max_obs=None, lazy_rollouts=False):
        self.tensor_args = {'device': 'cpu', 'dtype': torch.float32}
        self.q0 = torch.as_tensor(_np(q0))
        self.n_dof = self.q0.shape[0]
        self.DS_idx = 0
        self.DS_ARRAY = DS_ARRAY
        self.DS = DS_ARRAY[self.DS_idx]
        self.qf = self.DS.q_goal.squeeze()
        self.dh_params = torch.as_tensor(_np(dh_params))
        self.obs = torch.as_tensor(_np(obs)).reshape(-1, 4)
        self.n_obs = self.obs.shape[0]
        self.dt = dt
        self.dt_H = dt_H
        self.N_traj = N_traj
        self.dh_a = dh_a
        self.nn_model = nn_model
        self.n_closest_obs = n_closest_obs
        self.q_cur = self.q0
        self.policy_upd_rate = 0.1                      # MPPI.py:58 (the rate that is actually used, :344)
        self.dst_thr = 0.5
        self.ker_thr = 1e-3
        self.ignored_links = [0, 1, 2] if self.n_dof >= 7 else []
        self._device = device
        self._max_obs = int(max_obs or max(64, 2 * self.n_obs))
        self._engine = Engine(self.n_dof, N_traj, dt_H, n_closest_obs, self._max_obs, device=device)
        self._engine.set_mlp(nn_model.model.W, nn_model.model.b, nn_model.model.act, skip_after=getattr(nn_model.model, 'skip_after', ()))
        self._engine.set_obstacles(self.obs.numpy())
        self.Policy = TensorPolicyMPPI(N_traj, self.n_dof, self.tensor_args, engine=self._engine, seed=seed,
                                       rollout_offset=rollout_offset)
        self.Policy._owner = self
        self.Cost = Cost(self.qf, self.dh_params, owner=self)
        self.cur_cost = None
        self._cache = {}
        self._generation = 0          # propagate() calls so far (LazyRollout tensors belong to one of them)
        # False (default): propagate() returns and binds torch CPU tensors like the reference (one fetch of all rollout tensors per
        # call, 0.4 ms at 1024 x 32).  True: LazyRollout objects that stay on the GPU until read (lazy.py) -- for loops that, like
        # the reference's planner, read a handful of rows per iteration and never keep a tensor across propagate() calls
        self.lazy_rollouts = bool(lazy_rollouts)
        self._pushed = None           # what the device context last received (_push skips an unchanged set)
        self._pushed_version = -1     # Engine.config_version after that push
        self._qdot_cache = {}         # get_qdot values of the current cost (filled by shift_policy_means)
        self.all_traj = torch.zeros(N_traj, dt_H, self.n_dof)
        self.closest_dist_all = 100 + torch.zeros(N_traj, dt_H)
        self.qdot = torch.zeros(N_traj, self.n_dof)
        if warmup:                                       # MPPI.py:69-73 (optional here)
            for _ in range(5):
                self.Policy.sample_policy()
                self.propagate()

    # ---- DS switching (MPPI.py:75-84) -------------------------------------------------------------
    def reset_DS(self, DS):
        self.DS = DS
        self.qf = DS.q_goal.squeeze()
        self.Cost = Cost(self.qf, self.dh_params, owner=self)

    def switch_DS_idx(self, idx):
        self.DS_idx = idx
        self.DS = self.DS_ARRAY[idx]
        self.qf = self.DS.q_goal.squeeze()
        self.Cost = Cost(self.qf, self.dh_params, owner=self)

    def update_obstacles(self, obs):
        """MPPI.py:347-350."""
        self.obs = torch.as_tensor(_np(obs)).reshape(-1, 4)
        self.n_obs = self.obs.shape[0]
        # the reference takes any obstacle count at any time: the context grows its own obstacle buffers (omds_set_obstacles),
        # so the handle -- network, samples, RCCL communicator, screening state -- survives
        self._engine.set_obstacles(self.obs.numpy())
        self._max_obs = self._engine.max_obs
        return 0

    # ---- parameters -> device ----------------------------------------------------------------------
    def _push(self):
        """The mutable attributes the reference's callers poke (dt, dst_thr, ignored_links, Policy.p, DS, Cost limits) -> the
        device context; skipped while nothing changed since the last call (three ctypes calls per use otherwise)."""
        e = self._engine
        mask = 0
        for l in self.ignored_links:
            mask |= 1 << int(l)
        qf, dh, qmin, qmax = _np(self.qf), _np(self.dh_params), _np(self.Cost.q_min), _np(self.Cost.q_max)
        # the nominal DS by VALUE (an id() can be reused by a new object, and a SEDS DS can be edited in place)
        seds = self.DS.device_params() if hasattr(self.DS, "device_params") else None
        ds_sig = (type(self.DS).__name__, float(getattr(self.DS, "seds_thr", 0.0)), b"".join(a.tobytes() for a in seds) if seds else b"")
        sig = (float(self.dt), float(self.dst_thr), float(self.DS.lin_thr), float(self.Policy.p), mask, ds_sig, qf.tobytes(),
               dh.tobytes(), qmin.tobytes(), qmax.tobytes(), e.params.variant, e.params.cost_terms)
        # config_version: somebody configured the context through the Engine directly since our last push -> ours again
        if sig == self._pushed and e.config_version == self._pushed_version:
            return
        p = e.params
        p.dt, p.dst_thr, p.lin_thr, p.rbf_p, p.ignored_links = sig[0], sig[1], sig[2], sig[3], mask
        e.push_params()
        if seds is not None:                        # SEDS nominal DS (seds.py)
            e.set_ds_seds(qf, *seds, lin_thr=float(self.DS.lin_thr), seds_thr=float(self.DS.seds_thr))
        else:
            e.set_ds(qf)
        e.set_cost(dh, qmin, qmax)
        self._pushed = sig
        self._pushed_version = e.config_version

    # ---- rollouts (MPPI.py:97-224) -------------------------------------------------------------------
    def propagate(self, fetch=None):
        """MPPI.py:97-224.  Returns the reference's 5-tuple (all_traj, closest_dist_all, kernel_val_all[:, :, :K], dot_products,
        kernel_activations) and sets the same attributes.  By default they are torch CPU tensors, as in the reference (fresh ones
        every call, MPPI.py:86-91).  With ``lazy_rollouts = True`` (or ``fetch=False``) they are LazyRollout tensors (lazy.py): they
        stay on the GPU until read, single rollouts are fetched as rows, and the candidate search / cost / update work on the
        device-resident copies either way."""
        if fetch is None:
            fetch = not self.lazy_rollouts
        self._push()
        if self._engine.K != self.Policy.n_kernels:
            # like the reference, propagate() consumes whatever sample tensors exist; with a changed
            # kernel count and no sample_policy() call yet there is nothing valid to consume
            raise RuntimeError("Policy.n_kernels changed since the last sample_policy()/set_samples()")
        self._engine.propagate(_np(self.q_cur))
        self._cache = {}
        self.cur_cost = None
        self._qdot_cache = {}
        self._generation += 1
        N, H, n, K = self.N_traj, self.dt_H, self.n_dof, self.Policy.n_kernels
        if fetch:
            c = self._fetch()
            self.all_traj, self.closest_dist_all, self.kernel_val_all = c["all_traj"], c["closest_dist_all"], c["kernel_val_all"]
            self.dot_products, self.kernel_activations, self.qdot, self.normal_dirs = c["dot_products"], c["kernel_activations"], c["qdot"], c["normal"]
            # torch counts in-place writes: a caller who edits a returned tensor WITH TORCH OPERATIONS gets the host path of
            # check_traj_for_kernels.  (A write through a numpy view of the tensor -- t.numpy()[...] = x, np.asarray(t) -- shares its memory
            # without bumping t._version and is NOT seen: pass an edited copy instead.)
            self._versions = {id(t): t._version for t in c.values()}
        else:
            self._versions = {}
            lz = lambda key, shape: LazyRollout(self, key, shape, self._generation)
            self.all_traj = lz("all_traj", (N, H, n))
            self.closest_dist_all = lz("closest_dist_all", (N, H))
            self.kernel_val_all = lz("kernel_val_all", (N, H, K))
            self.dot_products = lz("dot_products", (N, H))
            self.kernel_activations = lz("kernel_activations", (N, H))
            self.qdot = lz("qdot", (N, n))
            self.normal_dirs = lz("normal", (N, H, n))      # norm_basis[..., 0]
        return (self.all_traj, self.closest_dist_all, self.kernel_val_all, self.dot_products, self.kernel_activations)

    def _is_device_copy(self, t):
        """True while ``t`` is a rollout tensor of the last propagate() that still equals the device-resident copy: a LazyRollout of
        this generation, or a fetched torch tensor nobody has written into since."""
        if isinstance(t, LazyRollout):
            return t._owner is self and t._gen == self._generation and not t._dirty
        return isinstance(t, torch.Tensor) and getattr(self, "_versions", {}).get(id(t), -1) == t._version

    def _fetch(self):
        """All rollout tensors of the last propagate as torch CPU tensors (one omds_get_rollouts), cached until the next one."""
        if not self._cache:
            self._cache = {k: torch.from_numpy(v) for k, v in self._engine.get_rollouts().items()}
        return self._cache

    @property
    def ker_w(self):
        kv = self._fetch()["kernel_val_all"]
        return kv[:, -1:, :].transpose(1, 2) if kv.numel() else None

    @property
    def norm_basis(self):
        """[N, H, n, n] basis whose column 0 is the obstacle normal (MPPI.py:122-127).  Column 0 comes
        from the device (normal_dirs); the tangent completion is the same LAPACK QR the reference calls
        (geqrf/orgqr via numpy), evaluated lazily and only for the entries a caller indexes -- nothing
        on the rollout path reads it (M v uses the closed form), the drivers read ONE entry per
        iteration (frankaPlanner.py:162)."""
        return _LazyBasis(self.normal_dirs)

    # ---- distance + gradient on arbitrary states (MPPI.py:227-282) ------------------------------------
    def distance_repulsion_nn(self, q_prev, aot=False):
        self._push()
        q = _np(q_prev).reshape(-1, self.n_dof)
        out_d, out_g = [], []
        for s in range(0, q.shape[0], self.N_traj):
            d, g, _, _ = self._engine.dist_grad(q[s:s + self.N_traj])
            out_d.append(d)
            out_g.append(g)
        self.nn_grad = torch.from_numpy(np.concatenate(out_g))
        return torch.from_numpy(np.concatenate(out_d)), self.nn_grad

    def update_kernel_normal_bases(self):
        """MPPI.py:284-304: re-evaluate the obstacle normal at every kernel centre."""
        K = self.Policy.n_kernels
        if K > 0:
            _, grad = self.distance_repulsion_nn(self.Policy.mu_c[0:K])
            g = grad.numpy()
            A = np.tile(np.eye(self.n_dof, dtype=np.float32), (K, 1, 1))
            A[:, :, 0] = g
            Q, _ = np.linalg.qr(A)
            Q = Q.astype(np.float32)
            Q[:, :, 0] = g / np.linalg.norm(g, axis=1, keepdims=True)
            self.Policy.kernel_obstacle_bases[0:K] = torch.from_numpy(Q)
        return 0

    # ---- cost and update (MPPI.py:315-345) -------------------------------------------------------------
    def get_cost(self):
        self._push()
        self.cur_cost = torch.from_numpy(self._engine.cost())
        self._qdot_cache = {}
        return self.cur_cost

    def init_comm(self, group=None):
        """Multi-GPU (one process per GPU, this object holds one shard of the rollouts; pass ``rollout_offset = rank * N_traj``
        to the constructor): creates the library's RCCL communicator over the ranks of ``group`` (any torch.distributed
        backend, it only carries the 128-byte id).  From then on ``shift_policy_means`` / ``get_qdot`` reduce over ALL shards
        on the device (csrc/comm.hip).  Raises OmdsError if RCCL is unusable -- there is no fallback."""
        from .dist import init_native_comm
        return init_native_comm(self._engine, group)

    def get_qdot(self, mode='best'):
        """MPPI.py:319-329.  After ``shift_policy_means()`` this is a local read, as in the reference: the update's reduction already
        produced both velocities (over ALL shards when a communicator exists) and they are kept until the next propagate / cost.
        Called BEFORE ``shift_policy_means()`` on a sharded context it has to run that reduction itself, which is a COLLECTIVE:
        then every rank must call it (a driver in which only rank 0 asks for the velocity would hang inside RCCL)."""
        if self.cur_cost is None:
            self.get_cost()
        if mode in self._qdot_cache:
            return self._qdot_cache[mode].clone()
        if self._engine.comm_info()[1] > 1:      # sharded: the arg-min / weighted mean over the rollouts of every rank (collective)
            K = self.Policy.n_kernels
            P = self.Policy
            _, _, _, _, qw, qb, _ = self._engine.weighted_update_sharded(0.0, self.ker_thr, P.mu_c.numpy()[:K], P.sigma_c.numpy()[:K],
                                                                         P.alpha_c.numpy()[:K], want_best=True)
            self._qdot_cache = {'weighted': torch.from_numpy(qw), 'best': torch.from_numpy(qb)}
            return self._qdot_cache[mode].clone()
        return torch.from_numpy(self._engine.get_qdot(mode))

    def shift_policy_means(self):
        """MPPI.py:331-345.  With a communicator (``init_comm``) the sums run over the rollouts of every rank: two all-reduces
        on the context stream inside ``omds_weighted_update_sharded``; without one the same call is the single-shard update."""
        if self.cur_cost is None:
            self.get_cost()
        P = self.Policy
        K = P.n_kernels
        mu, sg, al, mask, qw, qb, _ = self._engine.weighted_update_sharded(self.policy_upd_rate, self.ker_thr, P.mu_c.numpy(),
                                                                           P.sigma_c.numpy(), P.alpha_c.numpy(), want_best=True)
        if K > 0:
            P.mu_c[:K] = torch.from_numpy(mu)
            P.sigma_c[:K] = torch.from_numpy(sg)
            P.alpha_c[:K] = torch.from_numpy(al)
        self.update_mask = torch.from_numpy(mask)
        self.qdot_weighted = torch.from_numpy(qw)
        # get_qdot() of this cost is a local read from here on (the reduction above is the same one, over every shard)
        self._qdot_cache = {'weighted': torch.from_numpy(qw.copy()), 'best': torch.from_numpy(qb.copy())}
        return 0, int(mask.sum())

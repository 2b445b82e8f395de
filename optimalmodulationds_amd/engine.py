"""Thin object wrapper over one ``omds_ctx`` (one per GPU): numpy in, numpy out.

The reference-shaped classes (MPPI, TensorPolicyMPPI, ...) and bench.py sit on top of this."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class Engine:
    def __init__(self, n_dof, n_traj, horizon, n_closest, max_obs, n_kernel_max=50, device=0, flags=0, lib=None):
        self.lib = lib or L.load()   # lib: another build of the library (tests: _lib.load_test_hooks())
        self.n, self.N, self.H, self.k = int(n_dof), int(n_traj), int(horizon), int(n_closest)
        self.max_obs, self.Kmax, self.device = int(max_obs), int(n_kernel_max), int(device)
        cfg = L.OmdsConfig(self.n, self.N, self.H, self.Kmax, self.max_obs, self.k, self.device, int(flags))
        h = C.c_void_p()
        rc = self.lib.omds_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise L.OmdsError(f"omds_create failed ({rc}): {self.lib.omds_last_error(None).decode()}")
        self.h = h
        self.params = L.default_params()
        self.K = 0
        self.n_obs = 0
        self.C = 0
        self.config_version = 0      # bumped by every call that changes params / nominal DS / cost on the context (MPPI._push re-syncs on it)

    def close(self):
        if getattr(self, "h", None):
            self.lib.omds_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        L.check(self.h, rc, self.lib)

    # ---- configuration ------------------------------------------------------------------------
    def set_mlp(self, weights, biases, act="relu", out_div=None, skip_after=()):
        """``skip_after``: indices of the Linear layers behind whose activations the encoded input is concatenated
        (MLPRegression skips, network_macros_mod.py:142-146); empty for the plain sequential network."""
        Ws = [L.f32(w) for w in weights]
        bs = [L.f32(b) for b in biases]
        ins = np.array([w.shape[1] for w in Ws], dtype=np.int32)
        outs = np.array([w.shape[0] for w in Ws], dtype=np.int32)
        nl = len(Ws)
        Wp = (L.F32P * nl)(*[L.fptr(w) for w in Ws])
        bp = (L.F32P * nl)(*[L.fptr(b) for b in bs])
        C_out = int(outs[-1])
        if out_div is None:
            out_div = 100.0 if C_out == 9 else 1.0      # MPPI.py:236-237
        a = 0 if act == "relu" else 1
        sk = np.asarray(list(skip_after), dtype=np.int32)
        if sk.size or (ins[1:] != outs[:-1]).any():   # the explicit form also validates every layer's input width
            self._ck(self.lib.omds_set_mlp_ex(self.h, nl, L.iptr(ins), L.iptr(outs), Wp, bp, a, float(out_div), int(sk.size), L.iptr(sk)))
        else:
            dims = np.concatenate((ins[:1], outs)).astype(np.int32)
            self._ck(self.lib.omds_set_mlp(self.h, nl, L.iptr(dims), Wp, bp, a, float(out_div)))
        # a rejected network leaves the previous one installed (capi.hip), so the wrapper's view changes only on success
        self.C = C_out
        self.n_hidden_levels = nl - 1
        self.d = int(ins[0]) // 3                       # raw network inputs: n + 3, or n + 2 for the toy networks

    def set_obstacles(self, obs):
        """Any obstacle count: beyond ``max_obs`` the context grows its obstacle buffers (omds.h)."""
        obs = L.f32(obs).reshape(-1, 4)
        self._ck(self.lib.omds_set_obstacles(self.h, L.fptr(obs), obs.shape[0]))
        self.n_obs = obs.shape[0]
        self.max_obs = max(self.max_obs, self.n_obs)

    def set_ds(self, q_goal):
        q = L.f32(q_goal).reshape(self.n)
        self._ck(self.lib.omds_set_ds(self.h, L.fptr(q)))
        self.config_version += 1

    def set_ds_matrix(self, q_goal, A):
        """MPPI_toy's nominal DS: velocity = (q - q_goal) @ A (MPPI_toy.py:89)."""
        q = L.f32(q_goal).reshape(self.n)
        a = L.f32(A).reshape(self.n, self.n)
        self._ck(self.lib.omds_set_ds_matrix(self.h, L.fptr(q), L.fptr(a)))
        self.config_version += 1

    def set_ds_seds(self, q_goal, mu_in, b, sigma_inv, A, prior, den, lin_thr=1e-2, seds_thr=1e-2):
        """SEDS nominal DS (SEDS.py): G components; mu_in/b [G,n], sigma_inv/A [G,n,n], prior/den [G] as SEDS.__init__ / GMR
        derive them from the .mat file (optimalmodulationds_amd.seds.SEDS.device_params)."""
        q = L.f32(q_goal).reshape(self.n)
        mu_in, b = L.f32(mu_in), L.f32(b)
        G = mu_in.shape[0]
        si, a = L.f32(sigma_inv).reshape(G, self.n, self.n), L.f32(A).reshape(G, self.n, self.n)
        pr, dn = L.f32(prior).reshape(G), L.f32(den).reshape(G)
        self._ck(self.lib.omds_set_ds_seds(self.h, L.fptr(q), G, L.fptr(mu_in), L.fptr(b), L.fptr(si), L.fptr(a), L.fptr(pr), L.fptr(dn),
                                           float(lin_thr), float(seds_thr)))
        self.config_version += 1

    def push_params(self):
        self._ck(self.lib.omds_set_params(self.h, C.byref(self.params)))
        self.config_version += 1

    def set_cost(self, dh_params, q_min, q_max):
        dh = L.f32(dh_params).reshape(self.n + 1, 4)
        lo, hi = L.f32(q_min).reshape(self.n), L.f32(q_max).reshape(self.n)
        self._ck(self.lib.omds_set_cost(self.h, L.fptr(dh), L.fptr(lo), L.fptr(hi)))
        self.config_version += 1

    # ---- policy samples -----------------------------------------------------------------------
    def set_policy_samples(self, mu, sigma, alpha):
        mu = L.f32(mu)
        K = mu.shape[1] if mu.ndim == 3 else 0
        if K == 0:
            self._ck(self.lib.omds_set_policy_samples(self.h, None, None, None, 0))
        else:
            mu = mu.reshape(self.N, K, self.n)
            sg = L.f32(sigma).reshape(self.N, K)
            al = L.f32(alpha).reshape(self.N, K, self.n)
            self._ck(self.lib.omds_set_policy_samples(self.h, L.fptr(mu), L.fptr(sg), L.fptr(al), K))
        self.K = K

    def sample_policy(self, mu_c, sigma_c, alpha_c, mu_s, sigma_s, alpha_s, K, seed, rollout_offset=0):
        K = int(K)
        if K == 0:
            self._ck(self.lib.omds_sample_policy(self.h, None, None, None, 0, 0, 0, 0, int(seed), int(rollout_offset)))
        else:
            mu = L.f32(mu_c)[:K].reshape(K, self.n)
            sg = L.f32(sigma_c)[:K].reshape(K)
            al = L.f32(alpha_c)[:K].reshape(K, self.n)
            mu, sg, al = (np.ascontiguousarray(x) for x in (mu, sg, al))
            self._ck(self.lib.omds_sample_policy(self.h, L.fptr(mu), L.fptr(sg), L.fptr(al), float(mu_s), float(sigma_s),
                                                 float(alpha_s), K, int(seed), int(rollout_offset)))
        self.K = K

    def get_policy_samples(self):
        K = self.K
        mu = np.zeros((self.N, K, self.n), np.float32)
        sg = np.zeros((self.N, K), np.float32)
        al = np.zeros((self.N, K, self.n), np.float32)
        if K:
            self._ck(self.lib.omds_get_policy_samples(self.h, L.fptr(mu), L.fptr(sg), L.fptr(al)))
        return mu, sg, al

    # ---- rollouts -----------------------------------------------------------------------------
    def propagate(self, q_cur):
        q = L.f32(q_cur)
        per = 1 if q.ndim == 2 else 0
        q = q.reshape(self.N, self.n) if per else q.reshape(self.n)
        self._ck(self.lib.omds_propagate(self.h, L.fptr(q), per))

    def get_rollouts(self, want=("all_traj", "closest_dist_all", "kernel_val_all", "dot_products",
                                 "kernel_activations", "qdot", "normal")):
        N, H, n, K = self.N, self.H, self.n, self.K
        shapes = {"all_traj": (N, H, n), "closest_dist_all": (N, H), "kernel_val_all": (N, H, K),
                  "dot_products": (N, H), "kernel_activations": (N, H), "qdot": (N, n), "normal": (N, H, n)}
        out = {k: (np.zeros(shapes[k], np.float32) if k in want else None) for k in shapes}
        order = ["all_traj", "closest_dist_all", "kernel_val_all", "dot_products", "kernel_activations", "qdot", "normal"]
        ptrs = [L.fptr(out[k]) if (out[k] is not None and out[k].size) else None for k in order]
        self._ck(self.lib.omds_get_rollouts(self.h, *ptrs))
        return {k: v for k, v in out.items() if v is not None}

    def get_rollout_rows(self, t, want=("all_traj", "closest_dist_all", "kernel_val_all", "dot_products", "kernel_activations",
                                        "qdot", "normal")):
        """The same tensors for the rollouts ``t`` (indices) only: [len(t), H, ...] -- a few KB instead of the N x H tensors."""
        t = np.ascontiguousarray(np.atleast_1d(np.asarray(t)).astype(np.int32))
        R, H, n, K = t.shape[0], self.H, self.n, self.K
        shapes = {"all_traj": (R, H, n), "closest_dist_all": (R, H), "kernel_val_all": (R, H, K), "dot_products": (R, H),
                  "kernel_activations": (R, H), "qdot": (R, n), "normal": (R, H, n)}
        out = {k: (np.zeros(shapes[k], np.float32) if k in want else None) for k in shapes}
        order = ["all_traj", "closest_dist_all", "kernel_val_all", "dot_products", "kernel_activations", "qdot", "normal"]
        ptrs = [L.fptr(out[k]) if (out[k] is not None and out[k].size) else None for k in order]
        self._ck(self.lib.omds_get_rollout_rows(self.h, L.iptr(t), R, *ptrs))
        return {k: v for k, v in out.items() if v is not None}

    def dist_grad(self, q, want_mindist=False, want_idx=False):
        q = L.f32(q).reshape(-1, self.n)
        B = q.shape[0]
        dist = np.zeros(B, np.float32)
        grad = np.zeros((B, self.n), np.float32)
        mind = np.zeros((B, self.n_obs), np.float32) if want_mindist else None
        idx = np.zeros((B, self.k), np.int32) if want_idx else None
        self._ck(self.lib.omds_dist_grad(self.h, L.fptr(q), B, L.fptr(dist), L.fptr(grad), L.fptr(mind), L.iptr(idx)))
        return dist, grad, mind, idx

    def mlp_forward_vjp(self, x):
        x = L.f32(x).reshape(-1, self.d)
        B = x.shape[0]
        y = np.zeros((B, self.C), np.float32)
        g = np.zeros((B, self.d), np.float32)
        mi = np.zeros(B, np.int32)
        self._ck(self.lib.omds_mlp_forward_vjp(self.h, L.fptr(x), B, L.fptr(y), L.fptr(g), L.iptr(mi)))
        return y, g, mi

    def mlp_jacobian(self, x, cols):
        """(y [B,C], jac [B,d,len(cols)]): Jacobian columns of the listed raw outputs (omds_mlp_jacobian)."""
        x = L.f32(x).reshape(-1, self.d)
        cols = np.ascontiguousarray(cols, dtype=np.int32).reshape(-1)
        B = x.shape[0]
        y = np.zeros((B, self.C), np.float32)
        jac = np.zeros((B, self.d, cols.size), np.float32)
        self._ck(self.lib.omds_mlp_jacobian(self.h, L.fptr(x), B, L.iptr(cols), int(cols.size), L.fptr(y), L.fptr(jac)))
        return y, jac

    # ---- cost / update ------------------------------------------------------------------------
    def cost(self, fetch=True):
        c = np.zeros(self.N, np.float32) if fetch else None
        self._ck(self.lib.omds_cost(self.h, L.fptr(c)))
        return c

    def cost_eval(self, all_traj, closest_dist_all):
        """Cost.evaluate_costs on arbitrary [B,H,n] / [B,H] tensors (device evaluation, any B: chunks of N)."""
        tr = L.f32(all_traj).reshape(-1, self.H, self.n)
        di = L.f32(closest_dist_all).reshape(-1, self.H)
        out = np.zeros(tr.shape[0], np.float32)
        for s in range(0, tr.shape[0], self.N):
            a, b, o = np.ascontiguousarray(tr[s:s + self.N]), np.ascontiguousarray(di[s:s + self.N]), out[s:s + self.N]
            self._ck(self.lib.omds_cost_eval(self.h, L.fptr(a), L.fptr(b), a.shape[0], L.fptr(o)))
        return out

    def weighted_update(self, rate, ker_thr, mu_c, sigma_c, alpha_c, want_weights=False):
        K = self.K
        # copies: the C function updates the means in place and must not alias the caller's arrays
        mu = np.array(np.asarray(mu_c, dtype=np.float32)[:K], dtype=np.float32, order="C", copy=True).reshape(K, self.n)
        sg = np.array(np.asarray(sigma_c, dtype=np.float32)[:K], dtype=np.float32, order="C", copy=True).reshape(K)
        al = np.array(np.asarray(alpha_c, dtype=np.float32)[:K], dtype=np.float32, order="C", copy=True).reshape(K, self.n)
        mask = np.zeros(K, np.int32)
        w = np.zeros(self.N, np.float32) if want_weights else None
        self._ck(self.lib.omds_weighted_update(self.h, float(rate), float(ker_thr), L.fptr(mu), L.fptr(sg), L.fptr(al),
                                               L.iptr(mask), L.fptr(w)))
        return mu, sg, al, mask.astype(bool), w

    def weighted_update_eval(self, cost, kernel_val_all, kernel_activations, rate, ker_thr, mu_c, sigma_c, alpha_c, want_weights=False):
        """shift_policy_means on caller-supplied tensors (cost [N], kernel_val_all [N,H,K], kernel_activations [N,H]) against the
        policy samples the context holds; the context's own rollouts and cost stay untouched."""
        K = self.K
        c = L.f32(cost).reshape(self.N)
        kv = np.ascontiguousarray(L.f32(kernel_val_all).reshape(self.N, self.H, -1)[:, :, :K]) if K else None
        ka = L.f32(kernel_activations).reshape(self.N, self.H)
        mu = np.array(np.asarray(mu_c, dtype=np.float32)[:K], dtype=np.float32, order="C", copy=True).reshape(K, self.n)
        sg = np.array(np.asarray(sigma_c, dtype=np.float32)[:K], dtype=np.float32, order="C", copy=True).reshape(K)
        al = np.array(np.asarray(alpha_c, dtype=np.float32)[:K], dtype=np.float32, order="C", copy=True).reshape(K, self.n)
        mask = np.zeros(K, np.int32)
        w = np.zeros(self.N, np.float32) if want_weights else None
        self._ck(self.lib.omds_weighted_update_eval(self.h, L.fptr(c), L.fptr(kv), L.fptr(ka), float(rate), float(ker_thr),
                                                    L.fptr(mu), L.fptr(sg), L.fptr(al), L.iptr(mask), L.fptr(w)))
        return mu, sg, al, mask.astype(bool), w

    def get_qdot(self, mode="best"):
        out = np.zeros(self.n, np.float32)
        self._ck(self.lib.omds_get_qdot(self.h, 0 if mode == "best" else 1, L.fptr(out)))
        return out

    def cost_sum(self):
        out = np.zeros(2, np.float32)
        self._ck(self.lib.omds_cost_sum(self.h, L.fptr(out)))
        return out

    def local_sums(self, sum_cost, n_total, include_rollout0=True):
        red = np.zeros(self.lib.omds_red_count(self.h), np.float32)
        self._ck(self.lib.omds_local_sums(self.h, float(sum_cost), float(n_total), 1 if include_rollout0 else 0,
                                          L.fptr(red)))
        return red

    # ---- multi-GPU: native RCCL exchange (comm.hip) -------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte RCCL id (rank 0 makes it, the launcher ships it to the other ranks)."""
        lib = L.load()
        buf = (C.c_uint8 * 128)()
        rc = lib.omds_comm_unique_id(buf)
        if rc != 0:
            raise L.OmdsError(f"omds_comm_unique_id failed ({rc}): {(lib.omds_comm_last_error() or b'?').decode()}")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._ck(self.lib.omds_comm_init_rank(self.h, buf, int(rank), int(world)))

    def comm_destroy(self):
        self._ck(self.lib.omds_comm_destroy(self.h))

    def comm_info(self):
        r, w = C.c_int32(), C.c_int32()
        self._ck(self.lib.omds_comm_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_active(self):
        """True while the context owns an RCCL communicator (a single-rank one included)."""
        return bool(self.lib.omds_comm_active(self.h))

    def weighted_update_sharded(self, rate, ker_thr, mu_c, sigma_c, alpha_c, want_best=False):
        """MPPI.shift_policy_means + get_qdot over all shards of the communicator (all-reduces on the context
        stream, device buffers).  Returns (mu, sigma, alpha, mask, qdot_weighted, qdot_best | None, n_total)."""
        K = self.K
        mu = np.array(np.asarray(mu_c, dtype=np.float32)[:K], dtype=np.float32, order="C", copy=True).reshape(K, self.n)
        sg = np.array(np.asarray(sigma_c, dtype=np.float32)[:K], dtype=np.float32, order="C", copy=True).reshape(K)
        al = np.array(np.asarray(alpha_c, dtype=np.float32)[:K], dtype=np.float32, order="C", copy=True).reshape(K, self.n)
        mask = np.zeros(K, np.int32)
        qw = np.zeros(self.n, np.float32)
        qb = np.zeros(self.n, np.float32) if want_best else None
        nt = C.c_float()
        self._ck(self.lib.omds_weighted_update_sharded(self.h, float(rate), float(ker_thr), L.fptr(mu), L.fptr(sg),
                                                       L.fptr(al), L.iptr(mask), L.fptr(qw), L.fptr(qb),
                                                       C.cast(C.byref(nt), L.F32P)))
        return mu, sg, al, mask.astype(bool), qw, qb, nt.value

    def kernel_candidates(self, thr_dist, thr_kernel, thr_dot, mu_c, sigma_c, K, cap=None):
        """Device-side TensorPolicyMPPI.check_traj_for_kernels: (cand_q [m,n], cand_th [m,2], total)."""
        K = int(K)
        cap = int(self.N * self.H if cap is None else cap)
        mu = np.array(np.asarray(mu_c, dtype=np.float32)[:K], dtype=np.float32, order="C").reshape(K, self.n)
        sg = np.array(np.asarray(sigma_c, dtype=np.float32)[:K], dtype=np.float32, order="C").reshape(K)
        q = np.zeros((cap, self.n), np.float32)
        th = np.zeros((cap, 2), np.int32)
        cnt = C.c_int32(0)
        self._ck(self.lib.omds_kernel_candidates(self.h, float(thr_dist), float(thr_kernel), float(thr_dot), L.fptr(mu),
                                                 L.fptr(sg), K, cap, L.fptr(q), L.iptr(th), C.byref(cnt)))
        m = min(cnt.value, cap)
        return q[:m], th[:m], cnt.value

    # ---- screening of pass 1 (fp16 + exact re-selection) ----------------------------------------
    def set_screening(self, mode=-1, eps=0.0):
        """mode: -1 auto, 0 off (fp32 pass 1), 1 on; eps > 0 fixes the error bound, 0 keeps bound and calibration
        (mode change only), < 0 discards the calibration (measured again at the next screened propagate)."""
        self._ck(self.lib.omds_set_screening(self.h, int(mode), float(eps)))

    def set_screening_audit(self, one_in=64):
        """Audit sample of the pairs the screened step does not re-evaluate: 1 in ``one_in`` (power of two), 0 = none."""
        self._ck(self.lib.omds_set_screening_audit(self.h, int(one_in)))

    def set_screening_sweep(self, every=32, all_steps=False):
        """Every ``every``-th screened propagate checks ALL pairs of its last horizon step in fp32 (0 = never); with
        ``all_steps`` of every horizon step (soak / qualification runs)."""
        self._ck(self.lib.omds_set_screening_sweep(self.h, int(every), 1 if all_steps else 0))

    def sweep_hist(self, reset=False):
        """What all sweeps so far have counted (omds.h: omds_screen_sweep_hist), as a dict: pair counts, the largest
        ``Da - D`` over the pairs that were NOT candidates, and its distribution in log2 bins (``pos`` / ``neg``: bin b holds
        2^(b-32) <= |x| < 2^(b-31)) and in 128 linear bins of ``(Da - D) / eps`` over [0, 1) (``ratio``)."""
        w = (C.c_uint64 * L.SWEEP_HIST_WORDS)()
        self._ck(self.lib.omds_screen_sweep_hist(self.h, w, L.SWEEP_HIST_WORDS, 1 if reset else 0))
        a = np.frombuffer(w, dtype=np.uint64).copy()
        f = lambda bits: float(np.array([bits & 0xffffffff], np.uint32).view(np.float32)[0])
        Lb, Rb = L.SWEEP_HIST_LOG_BINS, L.SWEEP_HIST_RATIO_BINS
        return dict(pairs=int(a[0]), non_candidates=int(a[1]), above_half_eps=int(a[2]), above_eps=int(a[3]), non_finite=int(a[4]),
                    max_pos=f(int(a[5])), max_abs=f(int(a[6])), steps=int(a[7]), pos=a[8:8 + Lb].astype(np.int64),
                    neg=a[8 + Lb:8 + 2 * Lb].astype(np.int64), ratio=a[8 + 2 * Lb:8 + 2 * Lb + Rb].astype(np.int64))

    def screen_debug_corrupt(self, what, index, value=0.0):
        """Test hook (include/omds_test.h; needs ``lib=_lib.load_test_hooks()``): 0 = zero a weight fragment of the fp16 pack,
        1 = shift an obstacle in the screening inputs."""
        self._ck(self.lib.omds_screen_debug_corrupt(self.h, int(what), int(index), float(value)))

    def debug_force_tile_rows(self, tail_sel_rows=0, tail_rows=0):
        """Test hook (include/omds_test.h; needs ``lib=_lib.load_test_hooks()``; wide to that library): tile shape of the tail
        kernels; 0 = the launcher's own choice again."""
        self._ck(self.lib.omds_debug_force_tile_rows(int(tail_sel_rows), int(tail_rows)))

    def screen_mindist(self, q):
        q = L.f32(q).reshape(-1, self.n)
        out = np.zeros((q.shape[0], self.n_obs), np.float32)
        self._ck(self.lib.omds_screen_mindist(self.h, L.fptr(q), q.shape[0], L.fptr(out)))
        return out

    def screen_stats(self):
        act, eps, err = C.c_int32(), C.c_float(), C.c_float()
        cand, fb = C.c_double(), C.c_int64()
        self._ck(self.lib.omds_screen_stats(self.h, C.byref(act), C.cast(C.byref(eps), L.F32P), C.cast(C.byref(err), L.F32P),
                                            C.byref(cand), C.byref(fb)))
        one_in, susp, aerr = C.c_int32(), C.c_int32(), C.c_float()
        arows, ncal = C.c_double(), C.c_int64()
        self._ck(self.lib.omds_screen_audit_stats(self.h, C.byref(one_in), C.byref(arows), C.cast(C.byref(aerr), L.F32P),
                                                  C.byref(susp), C.byref(ncal)))
        sw_every, sw_n, sw_err = C.c_int32(), C.c_int64(), C.c_float()
        self._ck(self.lib.omds_screen_sweep_stats(self.h, C.byref(sw_every), C.byref(sw_n), C.byref(sw_err)))
        fe, fs, fo, ns = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.lib.omds_screen_fallback_stats(self.h, C.byref(fe), C.byref(fs), C.byref(fo), C.byref(ns)))
        nre = C.c_int64()
        never = np.zeros(9, np.int32)
        self._ck(self.lib.omds_screen_order_stats(self.h, C.byref(nre), L.iptr(never), 9))
        return dict(unit_reorders=nre.value, units_never_fired=[int(v) for v in never[:max(1, getattr(self, "n_hidden_levels", 9))]],
                    fallbacks_by_error=fe.value, fallbacks_by_slack=fs.value, fallbacks_by_overflow=fo.value, suspensions=ns.value,
                    active=bool(act.value), eps=eps.value, max_err_seen=err.value, candidates_per_rollout_step=cand.value,
                    fallbacks=fb.value, audit_one_in=one_in.value, audit_rows_per_rollout_step=arows.value,
                    audit_max_err=aerr.value, suspended=bool(susp.value), calibrations=ncal.value,
                    sweep_every=sw_every.value, sweeps=sw_n.value, sweep_max_err=sw_err.value)

    # ---- measurement --------------------------------------------------------------------------
    def pass1_skip_stats(self):
        """omds_pass1_skip_stats: dict(active, units, chunks [mean per tile and hidden level since set_mlp], tiles)."""
        act, tiles = C.c_int32(), C.c_int64()
        un, ch = (C.c_double * 9)(), (C.c_double * 9)()
        self._ck(self.lib.omds_pass1_skip_stats(self.h, C.byref(act), un, ch, 9, C.byref(tiles)))
        nl = getattr(self, "n_hidden_levels", 9)
        return dict(active=bool(act.value), units=[float(u) for u in un][:nl], chunks=[float(c) for c in ch][:nl], tiles=int(tiles.value))

    def sync(self):
        """Waits for everything enqueued on the context's stream (omds_sync)."""
        self._ck(self.lib.omds_sync(self.h))

    def prof_enable(self, on=True):
        """on: False/0 off, True/1 every launch of the dominant kernel, n > 1 every n-th launch."""
        self._ck(self.lib.omds_prof_enable(self.h, int(on)))

    def prof_reset(self):
        self._ck(self.lib.omds_prof_reset(self.h))

    def prof_read(self):
        ms, nl, nr = C.c_double(), C.c_int64(), C.c_int64()
        self._ck(self.lib.omds_prof_read(self.h, C.byref(ms), C.byref(nl), C.byref(nr)))
        return ms.value, nl.value, nr.value


def _prof_read_ex(self):
    ms, nl, fl, name = C.c_double(), C.c_int64(), C.c_double(), C.c_char_p()
    self._ck(self.lib.omds_prof_read_ex(self.h, C.byref(ms), C.byref(nl), C.byref(fl), C.byref(name)))
    return ms.value, nl.value, fl.value, (name.value or b"").decode()


Engine.prof_read_ex = _prof_read_ex


def red_layout(K, n):
    """Offsets into the packed reduction buffer (include/omds.h, omds_local_sums)."""
    o_mu = 1
    o_sg = o_mu + K * n
    o_al = o_sg + K
    o_mx = o_al + K * n
    o_ph = o_mx + K
    o_qd = o_ph + K
    o_best = o_qd + n
    return dict(sumw=0, mu=o_mu, sigma=o_sg, alpha=o_al, maxact=o_mx, phi0=o_ph, qdot=o_qd, best=o_best,
                n_sum=o_best, size=o_best + 1 + n)


def apply_update(K, n, H, red, n_total, rate, ker_thr, mu_c, sigma_c, alpha_c, variant=0):
    """Host arithmetic of the policy update on the (globally) reduced buffer -- omds_apply_update."""
    lib = L.load()
    red = L.f32(red)
    mu = np.array(np.asarray(mu_c, dtype=np.float32)[:K], dtype=np.float32, order="C", copy=True).reshape(K, n)
    sg = np.array(np.asarray(sigma_c, dtype=np.float32)[:K], dtype=np.float32, order="C", copy=True).reshape(K)
    al = np.array(np.asarray(alpha_c, dtype=np.float32)[:K], dtype=np.float32, order="C", copy=True).reshape(K, n)
    mask = np.zeros(K, np.int32)
    rc = lib.omds_apply_update(int(K), int(n), int(H), L.fptr(red), float(n_total), float(rate), float(ker_thr),
                               int(variant), L.fptr(mu), L.fptr(sg), L.fptr(al), L.iptr(mask))
    if rc != 0:
        raise L.OmdsError(f"omds_apply_update failed ({rc})")
    return mu, sg, al, mask.astype(bool)

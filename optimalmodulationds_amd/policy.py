"""RBF navigation-kernel policy -- mirrors ``ds_mppi/functions/policy.py``
(class TensorPolicyMPPI lines 12-175, eval_rbf :186-199, eval_rbf_simple :201-214).

The means (mu_c, sigma_c, alpha_c: a few hundred floats) live on the host like in the reference;
the N x K sampled tensors live on the GPU in SoA layout and are generated there
(``sample_policy`` -> omds_sample_policy, Philox + Box-Muller) -- ``mu_tmp / sigma_tmp /
alpha_tmp`` are fetched lazily in the reference layout when a caller reads them."""
from __future__ import annotations

import numpy as np
import torch


class TensorPolicyMPPI:
    def __init__(self, n_traj, n_dof, tensor_params=None, engine=None, seed=1234, rollout_offset=0):
        self.n_dof = n_dof
        self.n_traj = n_traj
        self.params = tensor_params or {}
        self.n_kernels = 0
        self.N_KERNEL_MAX = 50
        self.sigma_c_nominal = 0.2
        self.mu_c = torch.zeros((self.N_KERNEL_MAX, n_dof))
        self.sigma_c = torch.zeros((self.N_KERNEL_MAX,))
        self.alpha_c = torch.zeros((self.N_KERNEL_MAX, n_dof))
        self.mu_s = torch.tensor(0.0)
        self.sigma_s = torch.tensor(0.0)
        self.alpha_s = torch.tensor(0.0)
        self.q_min = torch.tensor([-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973])
        self.q_max = torch.tensor([2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973])
        self.rest = self.q_min + (self.q_max - self.q_min) * 0.5
        self.kernel_gammas = torch.zeros(self.N_KERNEL_MAX)
        self.kernel_obstacle_bases = torch.zeros((self.N_KERNEL_MAX, n_dof, n_dof))
        self.p = 2
        self.policy_upd_rate = 0.5   # written by the reference's drivers, read by nobody (MPPI.py:344)
        self._engine = engine
        self._seed = int(seed)
        self._draw = 0
        self._rollout_offset = int(rollout_offset)

    # ---- state -------------------------------------------------------------------------------
    def reset_policy(self):
        """policy.py:43-49: forget every kernel."""
        self.n_kernels = 0
        for t in (self.mu_c, self.sigma_c, self.alpha_c, self.kernel_gammas, self.kernel_obstacle_bases):
            t.zero_()

    def _need_engine(self):
        if self._engine is None:
            raise RuntimeError("this TensorPolicyMPPI is not attached to an MPPI/Engine (no device to sample on)")
        return self._engine

    def sample_policy(self):
        """policy.py:51-74 on the device; a fresh Philox key per call."""
        eng = self._need_engine()
        self._draw += 1
        eng.sample_policy(self.mu_c.numpy(), self.sigma_c.numpy(), self.alpha_c.numpy(), float(self.mu_s),
                          float(self.sigma_s), float(self.alpha_s), self.n_kernels,
                          seed=(self._seed << 20) + self._draw, rollout_offset=self._rollout_offset)

    def set_samples(self, mu_tmp, sigma_tmp, alpha_tmp):
        """Inject sampled tensors (first K kernels) instead of drawing them: parity runs, because the
        reference's torch-CPU RNG stream cannot be reproduced on the device."""
        self._need_engine().set_policy_samples(mu_tmp, sigma_tmp, alpha_tmp)

    def _tmp(self, which):
        mu, sg, al = self._need_engine().get_policy_samples()
        K = mu.shape[1]
        full = {"mu": np.zeros((self.n_traj, self.N_KERNEL_MAX, self.n_dof), np.float32),
                "sigma": np.zeros((self.n_traj, self.N_KERNEL_MAX), np.float32),
                "alpha": np.zeros((self.n_traj, self.N_KERNEL_MAX, self.n_dof), np.float32)}
        full["mu"][:, :K], full["sigma"][:, :K], full["alpha"][:, :K] = mu, sg, al
        return torch.from_numpy(full[which])

    mu_tmp = property(lambda self: self._tmp("mu"))
    sigma_tmp = property(lambda self: self._tmp("sigma"))
    alpha_tmp = property(lambda self: self._tmp("alpha"))

    def update_policy(self, w, upd_rate, update_mask=None):
        """policy.py:88-113 for caller-supplied weights.  Host convenience (fetches the samples);
        ``MPPI.shift_policy_means`` does NOT use it -- it reduces on the device."""
        K = self.n_kernels
        if K == 0:
            return
        mu, sg, al = self._need_engine().get_policy_samples()
        w = np.asarray(w, dtype=np.float32)
        upd = np.full(K, float(upd_rate), np.float32)
        if update_mask is not None:
            upd[~np.asarray(update_mask, dtype=bool)] = 0.0
        u = torch.from_numpy(upd)
        self.mu_c[:K] = (1 - u[:, None]) * self.mu_c[:K] + u[:, None] * torch.from_numpy((w[:, None, None] * mu).sum(0))
        self.sigma_c[:K] = (1 - u) * self.sigma_c[:K] + u * torch.from_numpy((w[:, None] * sg).sum(0))
        self.alpha_c[:K] = (1 - u[:, None]) * self.alpha_c[:K] + u[:, None] * torch.from_numpy((w[:, None, None] * al).sum(0))

    def update_with_data(self, data):
        """policy.py:115-127: install a policy dict received from the planner process (keys n_kernels, mu_c, alpha_c, sigma_c,
        norm_basis: payloads.planner_payload); rows past the received kernel count are cleared."""
        if data is None:
            return
        K = int(data["n_kernels"])
        for dst, key in ((self.mu_c, "mu_c"), (self.alpha_c, "alpha_c"), (self.sigma_c, "sigma_c"), (self.kernel_obstacle_bases, "norm_basis")):
            dst[:K] = torch.as_tensor(data[key], dtype=dst.dtype)
            dst[K:].zero_()
        self.n_kernels = K

    def add_kernel(self, q, kernel_gamma, kernel_obstacle_basis):
        """policy.py:129-151: a new kernel centred at q with the nominal width; its weights start as those of the nearest existing
        kernel (zero for the first one).  Host bookkeeping on the K x n means; a full policy is reported and left unchanged."""
        K = self.n_kernels
        if K >= self.N_KERNEL_MAX:
            print(f"kernel not added at {q}: the policy already holds N_KERNEL_MAX = {self.N_KERNEL_MAX} kernels")
            return
        q = torch.as_tensor(q, dtype=torch.float32).reshape(self.n_dof)
        if K:
            nearest = torch.linalg.vector_norm(self.mu_c[:K] - q, ord=2, dim=1).argmin()
            self.alpha_c[K] = self.alpha_c[nearest]
        else:
            self.alpha_c[K].zero_()
        self.mu_c[K] = q
        self.sigma_c[K] = float(self.sigma_c_nominal)
        self.kernel_gammas[K] = float(kernel_gamma)
        self.kernel_obstacle_bases[K] = torch.as_tensor(kernel_obstacle_basis, dtype=torch.float32)
        self.n_kernels = K + 1

    def check_traj_for_kernels(self, all_traj, closests_dist_all, dotproducts_all, thr_dist, thr_kernel, thr_dot):
        """policy.py:153-175: candidate kernel centres.  When the tensors are the owner MPPI's current
        rollouts (what every reference driver passes) the search runs on the device-resident copies
        (omds_kernel_candidates: flags + prefix sum + gather, reference order) and only the candidates
        cross PCIe; for foreign tensors the same logic runs on the host."""
        own = getattr(self, "_owner", None)
        if own is not None and all_traj is own.all_traj and closests_dist_all is own.closest_dist_all \
                and dotproducts_all is own.dot_products and self._engine is not None \
                and all(own._is_device_copy(t) for t in (all_traj, closests_dist_all, dotproducts_all)):
            own._push()     # Policy.p (the RBF norm order) and the other mutable attributes, when they changed
            q, th, total = self._engine.kernel_candidates(thr_dist, thr_kernel, thr_dot, self.mu_c.numpy(),
                                                          self.sigma_c.numpy(), self.n_kernels)
            self.last_candidate_index = torch.from_numpy(th.astype(np.int64))   # (t, h) of every candidate
            return torch.from_numpy(q)
        # foreign tensors: the same selection on the host
        states = torch.as_tensor(all_traj, dtype=torch.float32)
        near = (torch.as_tensor(closests_dist_all) < thr_dist) & (torch.as_tensor(dotproducts_all) < thr_dot)
        cand = states[near].reshape(-1, self.n_dof)
        K = self.n_kernels
        if K and cand.shape[0]:
            uncovered = eval_rbf_simple(cand, self.mu_c[:K], self.sigma_c[:K], self.p).amax(dim=-1) < thr_kernel   # NaN: not a candidate, like torch
            cand = cand[uncovered]
        return cand


def _rbf(q, mu, sigma, p):
    """exp(-sigma * ||q - mu||_p^2) with q [B, n] against mu [B, K, n] (per-row kernels) or [K, n] (shared kernels) -> [B, K]."""
    return torch.exp(-sigma * torch.linalg.vector_norm(q.unsqueeze(-2) - mu, ord=p, dim=-1) ** 2)


def eval_rbf(q, mu, sigma, p=2):
    """policy.py:186-199: per-rollout kernels mu [N, K, n], sigma [N, K] -> [N, K, 1] (host helper; the rollouts evaluate it inside
    the step kernels)."""
    return _rbf(q, mu, sigma, p).unsqueeze(2)


def eval_rbf_simple(q, mu, sigma, p=2):
    """policy.py:201-214: shared kernels mu [K, n], sigma [K] -> [B, K]."""
    return _rbf(q, mu, sigma, p)

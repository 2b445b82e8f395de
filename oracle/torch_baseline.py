"""CPU BASELINE for bench.py: the planner iteration in torch-CPU ops.  TEST / MEASUREMENT INFRASTRUCTURE ONLY.

The reference (epfl-lasa/OptimalModulationDS) runs this path as unfused PyTorch-CPU ops; its Python never travels to the
GPU box, so the timed CPU path is this restatement of the SAME op sequence: materialised
[N*O, n+4] network input, one ``addmm`` + ``relu`` per layer over all pairs, ``sort`` over the obstacles, forward + vjp of
the arg-min output on the N*k closest rows, softmax blend, per-rollout modulation, Euler step (FN/MPPI.py:97-282), then
``Cost.evaluate_costs`` (FN/cost.py:13-46, with the per-rollout FK loop of FN/fk_num.py:87-88 vectorised -- the reference's
TorchScript loop over N costs 2.4 ms per rollout and would dominate; the vectorised form is the faster, fairer baseline)
and ``shift_policy_means`` (FN/MPPI.py:331-345).  ``tests/test_oracle_golden.py`` pins it to the numpy oracle (which is
pinned to the reference's own outputs).  Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import it."""
from __future__ import annotations

import time

import numpy as np
import torch

from . import omds_oracle as orc


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))


class TorchPlanner:
    def __init__(self, m: orc.Mlp, obs, qf, dh_params, q_min, q_max, *, dt, k, ignored_links, prm: orc.Params):
        assert m.act == "relu"
        self.W = [_t(w) for w in m.W]
        self.b = [_t(b) for b in m.b]
        self.C = m.out_channels
        self.obs, self.qf = _t(obs), _t(qf)
        self.dh, self.q_min, self.q_max = _t(dh_params), _t(q_min), _t(q_max)
        self.dt, self.k, self.ign, self.prm = float(dt), int(k), list(ignored_links), prm
        self.goal_fk = self._link_endpoints(self.qf[None])[0]

    # ---- network (ML/network_macros_mod.py:137-146, ML/robot_sdf.py:153-158) -----------------------------------------
    def _forward(self, x, keep=False):
        h = torch.cat((x, torch.sin(x), torch.cos(x)), dim=-1)
        zs, hs = [], [h]
        for i in range(len(self.W) - 1):
            z = torch.addmm(self.b[i], h, self.W[i].t())
            h = torch.relu(z)
            if keep:
                zs.append(z)
                hs.append(h)
        y = torch.addmm(self.b[-1], h, self.W[-1].t())
        return (y, zs, hs) if keep else y

    def _vjp_argmin(self, x):
        d = x.shape[1]
        y, zs, hs = self._forward(x, keep=True)
        min_idx = torch.argmin(y, dim=1)
        g = self.W[-1][min_idx]
        for i in range(len(self.W) - 2, -1, -1):
            g = (g * (zs[i] > 0).to(g.dtype)) @ self.W[i]
        grad = g[:, :d] + g[:, d:2 * d] * torch.cos(x) - g[:, 2 * d:] * torch.sin(x)
        return y, grad, min_idx

    def distance_repulsion_nn(self, q):                       # FN/MPPI.py:227-282
        N, n = q.shape
        O, k = self.obs.shape[0], self.k
        nn_input = torch.hstack((q.tile(O, 1), self.obs.repeat_interleave(N, 0)))      # :93-95
        nn_dist = self._forward(nn_input[:, :-1])
        if self.C == 9:
            nn_dist = nn_dist / 100
        nn_dist = nn_dist - nn_input[:, -1:]
        if self.ign:
            nn_dist[:, self.ign] = 1e6
        mind = nn_dist.min(dim=1)[0].reshape(O, N).t()
        _, sort_idx = mind.sort(dim=1)
        sort_idx = sort_idx[:, :k]
        rows = (torch.arange(N)[:, None] + sort_idx * N).reshape(-1)
        nn_in2 = nn_input[rows]
        y, grad, min_idx = self._vjp_argmin(nn_in2[:, :-1])
        if self.C == 9:
            y = y / 100
        y = y - nn_in2[:, -1:]
        d = y[torch.arange(y.shape[0]), min_idx].reshape(N, k)
        g = grad[:, :n].reshape(N, k, n)
        w = torch.softmax(self.prm.softmax_k * d, dim=1)
        return d[:, 0].clone(), (g * w[:, :, None]).sum(dim=1)

    # ---- one horizon step after the network (FN/MPPI.py:102-217) --------------------------------------------------------
    @staticmethod
    def _gsig(x, y0, y1, x0, x1, kk):
        return y0 + (y1 - y0) / (1 + torch.exp(kk * (-x + (x0 + x1) / 2)))

    def modulation_step(self, q, d_raw, g_raw, mu, sg, al):
        p = self.prm
        x_dif = q - self.qf
        dst = torch.norm(x_dif, dim=-1)
        v = -x_dif
        far = dst > p.lin_thr
        v[far] = v[far] / dst[far][:, None]
        vnorm = torch.norm(v, dim=1, keepdim=True)
        vhat = v / vnorm
        distance = d_raw - p.dst_thr
        ghat = g_raw / torch.norm(g_raw, dim=1, keepdim=True)
        dot = (ghat * vhat).sum(-1)
        l_vel = self._gsig(dot, *p.lvel)
        l_n = self._gsig(distance, *p.ln)
        l_nv = l_vel + (1 - l_vel) * l_n
        l_tau = self._gsig(distance, *p.ltau)
        if mu.shape[1] > 0:
            nrm = torch.norm(q[:, None, :] - mu, p.p, dim=2)
            phi = torch.exp(-sg * nrm ** 2)
            pol = (al * phi[:, :, None]).sum(1)
        else:
            phi = q.new_zeros((q.shape[0], 0))
            pol = v * 0
        ga = (torch.sqrt(torch.abs(q - self.qf)).sum(1) ** 2).clamp(0, 1)
        ga[ga < p.goal_act_cut] = 0
        act = (1 - l_n) * (1 - l_vel) * ga
        v_tot = v + act[:, None] * pol * vnorm
        u = l_tau[:, None] * v_tot + ((l_nv - l_tau) * (ghat * v_tot).sum(1))[:, None] * ghat
        s = torch.norm(u, dim=1, keepdim=True)
        s[s <= p.norm_clamp] = 1
        u = torch.nan_to_num(u / s)
        coll = distance < 0
        u[coll] = u[coll] * p.coll_slow + (ghat * vnorm * p.coll_repulse)[coll]
        return u, distance, phi, act

    def propagate(self, q_cur, H, mu, sg, al):                # FN/MPPI.py:97-224
        N = mu.shape[0]
        n = self.qf.shape[0]
        K = mu.shape[1]
        all_traj = torch.zeros((N, H, n))
        dist_all = torch.zeros((N, H))
        kval_all = torch.zeros((N, H, K))
        acts = torch.zeros((N, H))
        all_traj[:, 0, :] = q_cur
        qdot = None
        for i in range(1, H + 1):
            q_prev = all_traj[:, i - 1, :].clone()
            d_raw, g_raw = self.distance_repulsion_nn(q_prev)
            u, dist, phi, act = self.modulation_step(q_prev, d_raw, g_raw, mu, sg, al)
            dist_all[:, i - 1] = dist
            acts[:, i - 1] = act
            kval_all[:, i - 1, :] = phi
            if i < H:
                all_traj[:, i, :] = q_prev + self.dt * u
            if i == 1:
                qdot = u.clone()
        return all_traj, dist_all, kval_all, acts, qdot

    # ---- cost (FN/cost.py:13-46) with a batched modified-DH chain (FN/fk_num.py:7-75) ---------------------------------
    def _link_endpoints(self, q):
        B, n = q.shape
        T = torch.eye(4).expand(B, 4, 4).clone()
        pts = []
        for i in range(n):
            dd, th, aa, alp = (float(v) for v in self.dh[i])
            sa, ca = float(np.sin(np.float32(alp))), float(np.cos(np.float32(alp)))
            sq, cq = torch.sin(q[:, i] + th), torch.cos(q[:, i] + th)
            M = torch.zeros((B, 4, 4))
            M[:, 0, 0], M[:, 0, 1], M[:, 0, 3] = cq, -sq, aa
            M[:, 1, 0], M[:, 1, 1], M[:, 1, 2], M[:, 1, 3] = sq * ca, cq * ca, -sa, -dd * sa
            M[:, 2, 0], M[:, 2, 1], M[:, 2, 2], M[:, 2, 3] = sq * sa, cq * sa, ca, dd * ca
            M[:, 3, 3] = 1
            T = T @ M
            pts.append(T[:, :3, 0] * float(self.dh[i + 1, 2]) + T[:, :3, 3])
        return torch.stack(pts, dim=1)

    def evaluate_costs(self, all_traj, dist_all):
        q_end = all_traj[:, -1, :]
        goal = 10 * torch.norm(q_end - self.qf, dim=1)
        coll = 100 * (dist_all < 0).sum(1)
        viol = ((all_traj < self.q_min).sum(1) + (all_traj > self.q_max).sum(1)).sum(1)
        jl = 100 * (viol > 0)
        stag = 10 * goal * torch.nan_to_num(1 / torch.norm(all_traj[:, 0, :] - q_end, dim=1))
        fk = 10 * torch.norm(self._link_endpoints(q_end) - self.goal_fk, dim=2).sum(1)
        return goal + coll + jl + stag + fk

    # ---- update (FN/MPPI.py:331-345, FN/policy.py:88-113) --------------------------------------------------------------
    @staticmethod
    def shift_policy_means(cost, kval_all, acts, mu_c, sg_c, al_c, mu, sg, al, rate, ker_thr):
        beta = cost.mean() / 50
        w = torch.exp(-1 / beta * cost)
        w = w / w.sum()
        m1 = (kval_all * acts[:, :, None]).max(dim=1)[0].mean(dim=0)
        m2 = kval_all[0].mean(dim=0)
        upd = rate * ((m1 > ker_thr) & (m2 > ker_thr)).to(mu_c.dtype)
        mu_n = (1 - upd)[:, None] * mu_c + upd[:, None] * (w[:, None, None] * mu).sum(0)
        sg_n = (1 - upd) * sg_c + upd * (w[:, None] * sg).sum(0)
        al_n = (1 - upd)[:, None] * al_c + upd[:, None] * (w[:, None, None] * al).sum(0)
        return mu_n, sg_n, al_n, w

    def iteration(self, q_cur, H, mu_c, sg_c, al_c, alpha_s, rate, ker_thr, gen):
        """sample_policy + propagate + get_cost + shift_policy_means (DS/frankaPlanner.py:132-145).  Returns the new means,
        qdot and the seconds spent in propagate / in everything else."""
        N = self._N
        K = mu_c.shape[0]
        t0 = time.perf_counter()
        mu = mu_c[None].expand(N, K, -1).clone()
        sg = sg_c[None].expand(N, K).clone()
        al = al_c[None] + alpha_s * torch.randn((N, K, mu_c.shape[1]), generator=gen)
        al[0] = al_c
        t1 = time.perf_counter()
        traj, dist, kval, acts, qdot = self.propagate(q_cur, H, mu, sg, al)
        t2 = time.perf_counter()
        cost = self.evaluate_costs(traj, dist)
        new = self.shift_policy_means(cost, kval, acts, mu_c, sg_c, al_c, mu, sg, al, rate, ker_thr)
        t3 = time.perf_counter()
        return new, qdot, t2 - t1, (t1 - t0) + (t3 - t2)


def time_iterations(planner: TorchPlanner, N, H, q0, mu_c, sg_c, al_c, alpha_s, ker_thr, threads, warmup=3, timed=5):
    """Medians over ``timed`` planner iterations at ``threads`` intra-op threads: (seconds in propagate, seconds in
    sample + cost + update)."""
    torch.set_num_threads(int(threads))
    planner._N = N
    gen = torch.Generator().manual_seed(1234)
    q = _t(q0)
    mc, sc, ac = _t(mu_c), _t(sg_c), _t(al_c)
    tp, tr = [], []
    with torch.no_grad():
        for it in range(warmup + timed):
            (mc, sc, ac, _), _, a, b = planner.iteration(q, H, mc, sc, ac, alpha_s, 0.1, ker_thr, gen)
            if it >= warmup:
                tp.append(a)
                tr.append(b)
    return float(np.median(tp)), float(np.median(tr))

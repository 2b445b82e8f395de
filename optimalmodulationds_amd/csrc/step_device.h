// Per-rollout device code shared by k_modulate (rollout_kernels.hip) and the fused per-step tail kernel
// (tail_kernel.hip): the whole of one horizon step after the distance network (MPPI.py:102-223).
#pragma once
#include "omds_internal.h"

// phase stamps inside modulate_core for the k_tail_sel timeline build (-DOMDS_TAIL_TL, mlp_device.h); nothing otherwise
#if defined(OMDS_TAIL_TL) && defined(OMDS_TL_STAMP)
#define OMDS_MOD_STAMP(i) OMDS_TL_STAMP(i)
#else
#define OMDS_MOD_STAMP(i)
#endif

#define FLT_MAX_F 3.402823466e+38f

__device__ __forceinline__ float nan_to_num_f(float x) {
    if (x != x) return 0.f;
    if (x == __builtin_inff()) return FLT_MAX_F;
    if (x == -__builtin_inff()) return -FLT_MAX_F;
    return x;
}

// generalized_sigmoid (MPPI.py:352-353); s = (y_min, y_max, x0, x1, k)
__device__ __forceinline__ float gsig(float x, const float* s) {
    const float c = (s[2] + s[3]) * 0.5f;
    return s[0] + (s[1] - s[0]) / (1.f + expf(s[4] * (-x + c)));
}

// One rollout t, handled by NSUB cooperating lanes (sub = 0..NSUB-1, consecutive lanes of one wave): every
// lane evaluates the scalar part redundantly, the loop over the K navigation kernels is strided over the
// lanes and its policy sum is combined with shuffles; lane 0 writes the per-rollout outputs.  gradx/drow
// hold the k gradient rows of this rollout starting at row grow0 (global memory or LDS).  q_in is the state
// all_traj[t, i-1]; q_next receives
// the integrated state (all lanes).
// WITH_SEDS: compile the SEDS nominal-DS branch in.  Only the stand-alone k_modulate does (omds_propagate routes contexts with a
// SEDS nominal DS through the step of stand-alone kernels): inlined into the fused step kernels the branch costs them 10-16
// registers and 1-3 % on every workload, out of line (a device function call) 226 registers and 20 %.
template <int ND, int NSUB, bool WITH_SEDS = false>
__device__ __forceinline__ void modulate_core(const StepArgs& a, const int i, int t, int sub, const float* gradx, const float* drow,
                                              int grow0, const float (&q_in)[ND], float (&q_next)[ND]) {
    const int N = a.N;
    const omds_params& p = a.prm;
    float q[ND], v[ND], vhat[ND], g[ND], vt[ND], u[ND], pol[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) q[j] = q_in[j];

    // nominal DS (LinDS.py:11-21) and its norm (MPPI.py:106-108)
    if (WITH_SEDS && a.seds != nullptr) {
        // SEDS.get_velocity (SEDS.py:34-74): Gaussian mixture regression on x = q - q_goal; the components are strided over
        // the NSUB lanes, their weighted outputs and the weight sum meet by shuffles.  The Mahalanobis forms are sums of terms
        // ~10^4 x their result, so the last bit of every product shows in the output: products and sums are rounded separately
        // (__fmul_rn / __fadd_rn, never fused), the arithmetic of the reference's elementwise multiply-then-sum and of the oracle
        const int st = omds_seds_stride(ND);
        float x[ND], ysum[ND], psum = 0.f, pj[4];   // up to 4 components per lane (G <= 64, NSUB = 16) or all of them (NSUB = 1, looped below)
        float dst2 = 0.f;
#pragma unroll
        for (int j = 0; j < ND; ++j) { x[j] = q[j] - a.qf[j]; dst2 = __fadd_rn(dst2, __fmul_rn(x[j], x[j])); ysum[j] = 0.f; }
        // pass 1: unnormalised responsibilities prior_j N_j(x) and their sum
        for (int gidx = sub, c = 0; gidx < a.seds_G; gidx += NSUB, ++c) {
            const float* g = a.seds + (size_t)gidx * st;
            const float* Si = g + 2 * ND + 2;
            float dd[ND], prob = 0.f;
#pragma unroll
            for (int j = 0; j < ND; ++j) dd[j] = x[j] - g[j];
#pragma unroll
            for (int cc = 0; cc < ND; ++cc) {      // prob = sum_c (sum_r dd_r Sinv[r][c]) dd_c   (SEDS.py:31)
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < ND; ++r) t = __fadd_rn(t, __fmul_rn(dd[r], Si[r * ND + cc]));
                prob = __fadd_rn(prob, __fmul_rn(t, dd[cc]));
            }
            const float pxi = g[2 * ND] * (expf(-0.5f * prob) / g[2 * ND + 1]);
            if (NSUB > 1 && c < 4) pj[c] = pxi;
            psum += pxi;
        }
        if (NSUB > 1) {
#pragma unroll
            for (int off = 1; off < NSUB; off <<= 1) psum += __shfl_xor(psum, off);
        }
        // pass 2: beta_j = clamp(nan_to_num(pxi / sum), 1e-8) (SEDS.py:49-51), y = sum_j beta_j (b_j + A_j (x - mu_j))
        for (int gidx = sub, c = 0; gidx < a.seds_G; gidx += NSUB, ++c) {
            const float* g = a.seds + (size_t)gidx * st;
            const float* Aj = g + 2 * ND + 2 + ND * ND;
            float pxi;
            if (NSUB > 1 && c < 4) pxi = pj[c];
            else {   // recompute (one lane per rollout, or more than four components per lane)
                const float* Si = g + 2 * ND + 2;
                float prob = 0.f;
#pragma unroll
                for (int cc = 0; cc < ND; ++cc) {
                    float t = 0.f;
#pragma unroll
                    for (int r = 0; r < ND; ++r) t = __fadd_rn(t, __fmul_rn(x[r] - g[r], Si[r * ND + cc]));
                    prob = __fadd_rn(prob, __fmul_rn(t, x[cc] - g[cc]));
                }
                pxi = g[2 * ND] * (expf(-0.5f * prob) / g[2 * ND + 1]);
            }
            float beta = nan_to_num_f(pxi / psum);
            beta = beta < 1e-8f ? 1e-8f : beta;
#pragma unroll
            for (int r = 0; r < ND; ++r) {
                float yj = g[ND + r];
#pragma unroll
                for (int cc = 0; cc < ND; ++cc) yj = __fadd_rn(yj, __fmul_rn(Aj[r * ND + cc], x[cc] - g[cc]));
                ysum[r] = __fadd_rn(ysum[r], __fmul_rn(beta, yj));
            }
        }
        if (NSUB > 1) {
#pragma unroll
            for (int j = 0; j < ND; ++j)
#pragma unroll
                for (int off = 1; off < NSUB; off <<= 1) ysum[j] += __shfl_xor(ysum[j], off);
        }
        float yn2 = 0.f;
#pragma unroll
        for (int j = 0; j < ND; ++j) yn2 = __fadd_rn(yn2, __fmul_rn(ysum[j], ysum[j]));
        const float dst = sqrtf(dst2), yn = sqrtf(yn2);
        const bool far = dst > a.seds_lin_thr, weak = yn < a.seds_thr;
#pragma unroll
        for (int j = 0; j < ND; ++j) v[j] = far ? (weak ? -x[j] / dst : ysum[j] / yn) : ysum[j];
    } else if (a.A == nullptr) {
        float dst2 = 0.f;
#pragma unroll
        for (int j = 0; j < ND; ++j) { const float xd = q[j] - a.qf[j]; v[j] = -xd; dst2 += xd * xd; }
        const float dst = sqrtf(dst2);
        if (dst > p.lin_thr) {
#pragma unroll
            for (int j = 0; j < ND; ++j) v[j] = v[j] / dst;
        }
    } else {   // MPPI_toy.py:89: (q - qf) @ A, not normalised
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < ND; ++r) acc += (q[r] - a.qf[r]) * a.A[r * ND + j];
            v[j] = acc;
        }
    }
    OMDS_MOD_STAMP(12);
    float vn2 = 0.f;
#pragma unroll
    for (int j = 0; j < ND; ++j) vn2 += v[j] * v[j];
    const float vnorm = sqrtf(vn2);
#pragma unroll
    for (int j = 0; j < ND; ++j) vhat[j] = v[j] / vnorm;

    // softmax(-10 d) blend of the k closest gradients; distance of the closest (MPPI.py:270-280)
    const int k = a.k, d = a.d;
    const float* dr = drow + grow0;
    float mx = -__builtin_inff();
    for (int jj = 0; jj < k; ++jj) mx = fmaxf(mx, p.softmax_k * dr[jj]);
    float ssum = 0.f;
    for (int jj = 0; jj < k; ++jj) ssum += expf(p.softmax_k * dr[jj] - mx);
#pragma unroll
    for (int j = 0; j < ND; ++j) g[j] = 0.f;
    for (int jj = 0; jj < k; ++jj) {
        const float w = expf(p.softmax_k * dr[jj] - mx) / ssum;
        const float* gr = gradx + (size_t)(grow0 + jj) * d;
#pragma unroll
        for (int j = 0; j < ND; ++j) g[j] += gr[j] * w;
    }
    OMDS_MOD_STAMP(13);
    const float distance = dr[0] - p.dst_thr;                               // MPPI.py:117
    if (sub == 0) a.distT[(size_t)(i - 1) * N + t] = distance;
    float gn2 = 0.f;
#pragma unroll
    for (int j = 0; j < ND; ++j) gn2 += g[j] * g[j];
    const float gn = sqrtf(gn2);
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        g[j] = g[j] / gn;                                                    // E[:, :, 0]  (MPPI.py:126)
        if (sub == 0) a.normalT[((size_t)(i - 1) * ND + j) * N + t] = g[j];
        dot += g[j] * vhat[j];
    }
    if (sub == 0) a.dotT[(size_t)(i - 1) * N + t] = dot;
    const float l_vel = gsig(dot, p.lvel);
    const float l_n = gsig(distance, p.ln);
    const float l_nv = l_vel * 1.f + (1.f - l_vel) * l_n;
    const float l_tau = gsig(distance, p.ltau);

    // activations (MPPI.py:190-196) -- they do not depend on the policy value
    const float ca = 1.f - l_n, va = 1.f - l_vel;
    float gs = 0.f;
#pragma unroll
    for (int j = 0; j < ND; ++j) gs += sqrtf(fabsf(q[j] - a.qf[j]));
    float ga = gs * gs;                                                     // norm(p=0.5)
    ga = ga < 0.f ? 0.f : (ga > 1.f ? 1.f : ga);
    if (ga < p.goal_act_cut) ga = 0.f;
    const float act = ca * va * ga;
    if (sub == 0) a.actT[(size_t)(i - 1) * N + t] = act;

    OMDS_MOD_STAMP(14);
    // RBF policy (policy.py:186-199, MPPI.py:165-186) + running statistics for the update mask
#pragma unroll
    for (int j = 0; j < ND; ++j) pol[j] = 0.f;
    for (int kk = sub; kk < a.K; kk += NSUB) {
        const float* mu = a.muT + (size_t)kk * ND * N + t;
        const float* al = a.alphaT + (size_t)kk * ND * N + t;
        float nrm;
        if (p.rbf_p == 2.f) {
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < ND; ++j) { const float df = q[j] - mu[(size_t)j * N]; s2 += df * df; }
            nrm = sqrtf(s2);
        } else {
            float sp = 0.f;
#pragma unroll
            for (int j = 0; j < ND; ++j) sp += powf(fabsf(q[j] - mu[(size_t)j * N]), p.rbf_p);
            nrm = powf(sp, 1.f / p.rbf_p);
        }
        const float phi = expf(-a.sigmaT[(size_t)kk * N + t] * (nrm * nrm));
        // MPPI_toy.py:178-179 multiplies the stored kernel values by the activation in place
        a.kvalT[((size_t)(i - 1) * a.Kmax + kk) * N + t] = (p.variant & OMDS_VARIANT_KVAL_TIMES_ACT) ? phi * act : phi;
#pragma unroll
        for (int j = 0; j < ND; ++j) pol[j] += al[(size_t)j * N] * phi;
        const float pa = phi * act;
        float* mp = a.maxact + (size_t)kk * N + t;
        if (i == 1) {
            *mp = pa;
        } else {
            const float old = *mp;
            *mp = (old != old || pa != pa) ? __builtin_nanf("") : fmaxf(old, pa);
        }
        if (t == 0) a.phisum0[kk] = (i == 1 ? 0.f : a.phisum0[kk]) + phi;
    }

    if (NSUB > 1) {
#pragma unroll
        for (int j = 0; j < ND; ++j)
#pragma unroll
            for (int off = 1; off < NSUB; off <<= 1) pol[j] += __shfl_xor(pol[j], off);
    }
    OMDS_MOD_STAMP(15);
    // total velocity, closed-form M v, normalisation, collision handling (MPPI.py:197-217)
    float gv = 0.f;
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const float pv = (act * pol[j]) * vnorm;
        vt[j] = v[j] + pv;
        gv += g[j] * vt[j];
    }
    float un2 = 0.f;
#pragma unroll
    for (int j = 0; j < ND; ++j) { u[j] = l_tau * vt[j] + ((l_nv - l_tau) * gv) * g[j]; un2 += u[j] * u[j]; }
    float s = sqrtf(un2);
    if (s <= p.norm_clamp) s = 1.f;
#pragma unroll
    for (int j = 0; j < ND; ++j) u[j] = nan_to_num_f(u[j] / s);
    if (distance < 0.f) {
#pragma unroll
        for (int j = 0; j < ND; ++j) { u[j] *= p.coll_slow; u[j] += (g[j] * vnorm) * p.coll_repulse; }
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) q_next[j] = q[j] + p.dt * u[j];
    if (sub == 0) {
        if (i < a.H) {
            float* qn = a.trajT + (size_t)i * ND * N;
#pragma unroll
            for (int j = 0; j < ND; ++j) qn[(size_t)j * N + t] = q_next[j];
        }
        if (i == 1) {
#pragma unroll
            for (int j = 0; j < ND; ++j) a.qdotT[(size_t)j * N + t] = u[j];
        }
    }
    OMDS_MOD_STAMP(16);
}


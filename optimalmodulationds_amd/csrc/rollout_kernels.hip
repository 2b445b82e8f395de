// Per-rollout kernels for gfx950: modulation + policy + Euler step, cost, MPPI reductions,
// device-side policy sampling, and layout conversions.
//
// What they replace in the reference (paths relative to python_scripts/ds_mppi/functions/):
//   k_modulate   : LinDS.get_velocity (LinDS.py:11-21), the softmax blend of the k gradients
//                  (MPPI.py:270-280), basis/eigenvalues/M (MPPI.py:120-161, closed form
//                  M v = l_tau v + (l_nv - l_tau)(g.v) g, no QR), eval_rbf + policy sum
//                  (policy.py:186-199, MPPI.py:165-186), activation / apply / collision handling
//                  (MPPI.py:187-217) and the Euler step (MPPI.py:218-223)
//   k_cost       : Cost.evaluate_costs incl. the modified-DH forward kinematics
//                  (cost.py:13-46, fk_num.py:7-89)
//   k_cost_sum, k_weights, k_policy_sums : MPPI.shift_policy_means / get_qdot /
//                  TensorPolicyMPPI.update_policy sums (MPPI.py:319-345, policy.py:88-113)
//   k_sample     : TensorPolicyMPPI.sample_policy (policy.py:51-74) with a counter-based RNG
//
// Layout: all per-rollout state is SoA with the rollout index fastest ([H][n][N], [K][n][N]), so
// one-thread-per-rollout kernels read and write fully coalesced; reference (AoS) layouts are
// produced on demand by the permute kernels at the bottom.
#include "omds_internal.h"

#include <cmath>

#include "step_device.h"

template <int ND>
__global__ __launch_bounds__(256) void k_modulate(StepArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.N) return;
    float q[ND], qn[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) q[j] = a.trajT[((size_t)(a.step - 1) * ND + j) * a.N + t];
    modulate_core<ND, 1, true>(a, a.step, t, 0, a.gradx, a.drow, t * a.k, q, qn);
}

template <int ND>
static void launch_modulate_t(hipStream_t s, const StepArgs& a) {
    hipLaunchKernelGGL(k_modulate<ND>, dim3((a.N + 255) / 256), dim3(256), 0, s, a);
}

void omds_launch_modulate(hipStream_t s, const StepArgs& a) {
    switch (a.n) {
        case 1: launch_modulate_t<1>(s, a); break;
        case 2: launch_modulate_t<2>(s, a); break;
        case 3: launch_modulate_t<3>(s, a); break;
        case 4: launch_modulate_t<4>(s, a); break;
        case 5: launch_modulate_t<5>(s, a); break;
        case 6: launch_modulate_t<6>(s, a); break;
        default: launch_modulate_t<7>(s, a); break;
    }
}

// ------------------------------------------------------------------------------------------------
// cost
// ------------------------------------------------------------------------------------------------
// Modified-DH chain (fk_num.py:7-47); pts[i] = frame_{i+1} applied to [a_{i+1}, 0, 0] (fk_num.py:63-73)
template <int ND>
__host__ __device__ inline void link_endpoints_t(const float* q, const float* dh, float* pts) {
    float T[3][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};
    for (int i = 0; i < ND; ++i) {
        const float dd = dh[i * 4 + 0], th = dh[i * 4 + 1], aa = dh[i * 4 + 2], al = dh[i * 4 + 3];
        const float sa = sinf(al), ca = cosf(al), sq = sinf(q[i] + th), cq = cosf(q[i] + th);
        const float M[4][4] = {{cq, -sq, 0.f, aa}, {sq * ca, cq * ca, -sa, -dd * sa}, {sq * sa, cq * sa, ca, dd * ca},
                               {0.f, 0.f, 0.f, 1.f}};
        float Tn[3][4];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                float acc = 0.f;
                for (int kk = 0; kk < 3; ++kk) acc += T[r][kk] * M[kk][c];
                acc += T[r][3] * M[3][c];
                Tn[r][c] = acc;
            }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) T[r][c] = Tn[r][c];
        const float a1 = dh[(i + 1) * 4 + 2];
        for (int r = 0; r < 3; ++r) pts[i * 3 + r] = T[r][0] * a1 + T[r][3];
    }
}

void omds_host_link_endpoints(const float* q, const float* dh, int n, float* pts) {
    switch (n) {
        case 1: link_endpoints_t<1>(q, dh, pts); break;
        case 2: link_endpoints_t<2>(q, dh, pts); break;
        case 3: link_endpoints_t<3>(q, dh, pts); break;
        case 4: link_endpoints_t<4>(q, dh, pts); break;
        case 5: link_endpoints_t<5>(q, dh, pts); break;
        case 6: link_endpoints_t<6>(q, dh, pts); break;
        default: link_endpoints_t<7>(q, dh, pts); break;
    }
}

template <int ND>
__global__ __launch_bounds__(256) void k_cost(CostArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.N, H = a.H;
    if (t >= N) return;
    float q0[ND], qe[ND];
    int viol = 0, ncoll = 0;
    for (int h = 0; h < H; ++h) {
        const float* qp = a.trajT + (size_t)h * ND * N + t;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const float x = qp[(size_t)j * N];
            if (h == 0) q0[j] = x;
            if (h == H - 1) qe[j] = x;
            viol |= (x < a.qmin[j]) | (x > a.qmax[j]);
        }
        ncoll += (a.distT[(size_t)h * N + t] < 0.f) ? 1 : 0;
    }
    float g2 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const float dg = qe[j] - a.qf[j], ds = q0[j] - qe[j];
        g2 += dg * dg;
        s2 += ds * ds;
    }
    const float goal = 10.f * sqrtf(g2);
    const float coll = 100.f * (float)ncoll;
    const float jl = viol ? 100.f : 0.f;
    const float stag = (10.f * goal) * nan_to_num_f(1.f / sqrtf(s2));
    float pts[ND * 3];
    link_endpoints_t<ND>(qe, a.dh, pts);
    float fk = 0.f;
#pragma unroll
    for (int l = 0; l < ND; ++l) {
        const float dx = pts[l * 3] - a.goal_fk[l * 3], dy = pts[l * 3 + 1] - a.goal_fk[l * 3 + 1],
                    dz = pts[l * 3 + 2] - a.goal_fk[l * 3 + 2];
        fk += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    // the reference sums left to right (cost.py:21); cost_toy.py:18 leaves out the joint-limit and FK terms
    float total = (a.terms & OMDS_COST_GOAL) ? goal : 0.f;
    if (a.terms & OMDS_COST_COLLISION) total += coll;
    if (a.terms & OMDS_COST_JOINT_LIMITS) total += jl;
    if (a.terms & OMDS_COST_STAGNATION) total += stag;
    if (a.terms & OMDS_COST_FK) total += 10.f * fk;
    a.cost[t] = total;
}

template <int ND>
static void launch_cost_t(hipStream_t s, const CostArgs& a) {
    hipLaunchKernelGGL(k_cost<ND>, dim3((a.N + 255) / 256), dim3(256), 0, s, a);
}

void omds_launch_cost(hipStream_t s, const CostArgs& a) {
    switch (a.n) {
        case 1: launch_cost_t<1>(s, a); break;
        case 2: launch_cost_t<2>(s, a); break;
        case 3: launch_cost_t<3>(s, a); break;
        case 4: launch_cost_t<4>(s, a); break;
        case 5: launch_cost_t<5>(s, a); break;
        case 6: launch_cost_t<6>(s, a); break;
        default: launch_cost_t<7>(s, a); break;
    }
}

// ------------------------------------------------------------------------------------------------
// reductions (deterministic: fixed per-thread strides + fixed LDS tree)
// ------------------------------------------------------------------------------------------------
constexpr int RED_NT = 256;

__device__ __forceinline__ float block_sum(float v, float* sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int s = RED_NT / 2; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const float r = sh[0];
    __syncthreads();
    return r;
}

// red2 = [sum(cost), N]
__global__ __launch_bounds__(RED_NT) void k_cost_sum(const float* __restrict__ cost, int N, float* __restrict__ red2) {
    __shared__ float sh[RED_NT];
    float s = 0.f;
    for (int t = threadIdx.x; t < N; t += RED_NT) s += cost[t];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) { red2[0] = s; red2[1] = (float)N; }
}

// w'[t] = exp(-1/beta * cost[t]), beta = mean(cost)/50 over ALL shards (MPPI.py:332-333)
__global__ __launch_bounds__(256) void k_weights(const float* __restrict__ cost, int N, const float* __restrict__ red2,
                                                 float* __restrict__ w) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    const float beta = (red2[0] / red2[1]) / 50.f;
    w[t] = expf((-1.f / beta) * cost[t]);
}

// Packed reduction buffer (floats), K = active kernels, n = dof:
//   [0]                      sum_t w'
//   [1 .. 1+K*n)             sum_t w' mu_tmp[t,kappa,:]
//   [.. +K)                  sum_t w' sigma_tmp[t,kappa]
//   [.. +K*n)                sum_t w' alpha_tmp[t,kappa,:]
//   [.. +K)                  sum_t max_h(phi*act)[t,kappa]          (MPPI.py:336-338)
//   [.. +K)                  sum_h phi[0,h,kappa] of GLOBAL rollout 0 (MPPI.py:341), else 0
//   [.. +n)                  sum_t w' qdot[t,:]                     (get_qdot 'weighted')
//   [.. +1+n)                min cost of this shard, qdot of its arg-min ('best'; NOT summable)
int omds_red_size(int K, int n) { return 1 + K * (2 * n + 3) + n + 1 + n; }

__global__ __launch_bounds__(RED_NT) void k_policy_sums(int N, int n, int K, const float* __restrict__ w,
                                                        const float* __restrict__ muT, const float* __restrict__ sigmaT,
                                                        const float* __restrict__ alphaT, const float* __restrict__ maxact,
                                                        const float* __restrict__ phisum0, const float* __restrict__ qdotT,
                                                        const float* __restrict__ cost, int include_rollout0,
                                                        float* __restrict__ red) {
    __shared__ float sh[RED_NT];
    __shared__ int shi[RED_NT];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int o_mu = 1, o_sg = o_mu + K * n, o_al = o_sg + K, o_mx = o_al + K * n, o_ph = o_mx + K, o_qd = o_ph + K,
              o_best = o_qd + n;
    // one sum per workgroup (blockIdx.y = which): the 2n + 2 sums of a kernel used to run one after the other in one workgroup,
    // 16 x 10 barriers = 20 us per iteration; each sum is the same strided accumulation and the same LDS tree as before
    const int part = blockIdx.y;
    if (b < K) {
        if (part < 2 * n) {
            const int j = part < n ? part : part - n;
            const float* src = (part < n ? muT : alphaT) + ((size_t)b * n + j) * N;
            float s = 0.f;
            for (int t = tid; t < N; t += RED_NT) s += w[t] * src[t];
            s = block_sum(s, sh);
            if (tid == 0) red[(part < n ? o_mu : o_al) + b * n + j] = s;
        } else if (part == 2 * n) {
            float s = 0.f;
            for (int t = tid; t < N; t += RED_NT) s += w[t] * sigmaT[(size_t)b * N + t];
            s = block_sum(s, sh);
            if (tid == 0) red[o_sg + b] = s;
        } else if (part == 2 * n + 1) {
            float s = 0.f;
            for (int t = tid; t < N; t += RED_NT) s += maxact[(size_t)b * N + t];
            s = block_sum(s, sh);
            if (tid == 0) {
                red[o_mx + b] = s;
                red[o_ph + b] = include_rollout0 ? phisum0[b] : 0.f;
            }
        }
    } else if (part == 0) {
        float s = 0.f;
        for (int t = tid; t < N; t += RED_NT) s += w[t];
        s = block_sum(s, sh);
        if (tid == 0) red[0] = s;
    } else if (part <= n) {
        const int j = part - 1;
        float s = 0.f;
        for (int t = tid; t < N; t += RED_NT) s += w[t] * qdotT[(size_t)j * N + t];
        s = block_sum(s, sh);
        if (tid == 0) red[o_qd + j] = s;
    } else if (part == n + 1) {
        // arg-min of the cost, first index on ties (torch.argmin on CPU)
        float bv = __builtin_inff();
        int bi = 0x7fffffff;
        for (int t = tid; t < N; t += RED_NT) {
            const float c = cost[t];
            if (c < bv) { bv = c; bi = t; }
        }
        sh[tid] = bv;
        shi[tid] = bi;
        __syncthreads();
        for (int st = RED_NT / 2; st > 0; st >>= 1) {
            if (tid < st) {
                const float ov = sh[tid + st];
                const int oi = shi[tid + st];
                if (ov < sh[tid] || (ov == sh[tid] && oi < shi[tid])) { sh[tid] = ov; shi[tid] = oi; }
            }
            __syncthreads();
        }
        if (tid == 0) {
            int best = shi[0];
            if (best == 0x7fffffff) best = 0;
            red[o_best] = sh[0];
            for (int j = 0; j < n; ++j) red[o_best + 1 + j] = qdotT[(size_t)j * N + best];
        }
    }
}

void omds_launch_cost_sum(hipStream_t s, const float* cost, int N, float* red2) {
    hipLaunchKernelGGL(k_cost_sum, dim3(1), dim3(RED_NT), 0, s, cost, N, red2);
}
void omds_launch_weights(hipStream_t s, const float* cost, int N, const float* red2_global, float* w, float*) {
    hipLaunchKernelGGL(k_weights, dim3((N + 255) / 256), dim3(256), 0, s, cost, N, red2_global, w);
}
void omds_launch_policy_sums(hipStream_t s, int N, int n, int K, const float* w, const float* muT, const float* sigmaT,
                             const float* alphaT, const float* maxact, const float* phisum0, const float* qdotT,
                             const float* cost, int include_rollout0, float* red) {
    hipLaunchKernelGGL(k_policy_sums, dim3(K + 1, 2 * n + 2), dim3(RED_NT), 0, s, N, n, K, w, muT, sigmaT, alphaT, maxact, phisum0,
                       qdotT, cost, include_rollout0, red);
}

// ------------------------------------------------------------------------------------------------
// policy sampling: Philox4x32-10 keyed by the seed, counter = (global rollout, kernel, draw)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t* out) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1,
                       n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float* z0, float* z1) {
    const float u1 = ((float)a + 1.f) * 2.3283064365386963e-10f;   // (0, 1]
    const float u2 = (float)b * 2.3283064365386963e-10f;           // [0, 1]
    const float r = sqrtf(-2.f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    *z0 = r * cs;
    *z1 = r * sn;
}

// means = [mu_c (K*n) | sigma_c (K) | alpha_c (K*n)]
__global__ __launch_bounds__(256) void k_sample(int N, int n, int K, const float* __restrict__ means, float mu_s,
                                                float sigma_s, float alpha_s, uint64_t seed, long long rollout_offset,
                                                float* __restrict__ muT, float* __restrict__ sigmaT,
                                                float* __restrict__ alphaT) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int kk = blockIdx.y;
    if (t >= N) return;
    const unsigned long long gt = (unsigned long long)(rollout_offset + t);
    float z[2 * OMDS_MAX_DOF + 2];
    const int need = 2 * n + 1;
    for (int c = 0; c * 4 < need + 1; ++c) {
        uint32_t r[4];
        philox4x32_10((uint32_t)gt, (uint32_t)(gt >> 32), (uint32_t)kk, (uint32_t)c, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        float a0, a1, b0, b1;
        box_muller(r[0], r[1], &a0, &a1);
        box_muller(r[2], r[3], &b0, &b1);
        const float v[4] = {a0, a1, b0, b1};
        for (int e = 0; e < 4; ++e)
            if (c * 4 + e < 2 * OMDS_MAX_DOF + 2) z[c * 4 + e] = v[e];
    }
    const float* mu_c = means + (size_t)kk * n;
    const float* sg_c = means + (size_t)K * n + kk;
    const float* al_c = means + (size_t)K * n + K + (size_t)kk * n;
    const bool mean_rollout = (gt == 0);   // policy.py:74: rollout 0 carries the un-noised alpha
    for (int j = 0; j < n; ++j) {
        muT[((size_t)kk * n + j) * N + t] = z[j] * mu_s + mu_c[j];
        alphaT[((size_t)kk * n + j) * N + t] = mean_rollout ? al_c[j] : (z[n + 1 + j] * alpha_s + al_c[j]);
    }
    sigmaT[(size_t)kk * N + t] = z[n] * sigma_s + sg_c[0];
}

void omds_launch_sample(hipStream_t s, int N, int n, int K, const float* means, float mu_s, float sigma_s, float alpha_s,
                        uint64_t seed, int64_t rollout_offset, float* muT, float* sigmaT, float* alphaT) {
    if (K <= 0) return;
    hipLaunchKernelGGL(k_sample, dim3((N + 255) / 256, K), dim3(256), 0, s, N, n, K, means, mu_s, sigma_s, alpha_s, seed,
                       (long long)rollout_offset, muT, sigmaT, alphaT);
}

// ------------------------------------------------------------------------------------------------
// navigation-kernel candidates (TensorPolicyMPPI.check_traj_for_kernels, policy.py:153-175) on the
// device-resident rollouts; output order = the reference's boolean-mask order (rollout, then horizon)
// ------------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(256) void k_cand_flags(int N, int H, int K, const float* __restrict__ trajT,
                                                    const float* __restrict__ distT, const float* __restrict__ dotT,
                                                    const float* __restrict__ means /* mu_c [K][n] | sigma_c [K] */,
                                                    float thr_dist, float thr_kernel, float thr_dot, float rbf_p,
                                                    unsigned char* __restrict__ flags, int* __restrict__ counts) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    int cnt = 0;
    for (int h = 0; h < H; ++h) {
        bool cand = (distT[(size_t)h * N + t] < thr_dist) && (dotT[(size_t)h * N + t] < thr_dot);
        if (cand && K > 0) {
            float q[ND];
#pragma unroll
            for (int j = 0; j < ND; ++j) q[j] = trajT[((size_t)h * ND + j) * N + t];
            float best = -__builtin_inff();
            for (int kk = 0; kk < K; ++kk) {
                float nrm;
                if (rbf_p == 2.f) {
                    float s2 = 0.f;
#pragma unroll
                    for (int j = 0; j < ND; ++j) { const float df = q[j] - means[kk * ND + j]; s2 += df * df; }
                    nrm = sqrtf(s2);
                } else {
                    float sp = 0.f;
#pragma unroll
                    for (int j = 0; j < ND; ++j) sp += powf(fabsf(q[j] - means[kk * ND + j]), rbf_p);
                    nrm = powf(sp, 1.f / rbf_p);
                }
                best = fmaxf(best, expf(-means[K * ND + kk] * (nrm * nrm)));
            }
            cand = best < thr_kernel;
        }
        flags[(size_t)t * H + h] = cand ? 1 : 0;
        cnt += cand ? 1 : 0;
    }
    counts[t] = cnt;
}

// exclusive scan of counts[N] -> offsets[N], total in offsets[N]; single block
__global__ __launch_bounds__(1024) void k_cand_scan(const int* __restrict__ counts, int N, int* __restrict__ offsets) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (N + 1023) / 1024;
    const int lo = tid * per, hi = min(N, lo + per);
    int s = 0;
    for (int t = lo; t < hi; ++t) s += counts[t];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int base = part[tid] - s;
    for (int t = lo; t < hi; ++t) { offsets[t] = base; base += counts[t]; }
    if (tid == 1023) offsets[N] = part[1023];
}

__global__ __launch_bounds__(256) void k_cand_write(int N, int H, int n, const float* __restrict__ trajT,
                                                    const unsigned char* __restrict__ flags, const int* __restrict__ offsets,
                                                    int cap, float* __restrict__ cand_q, int* __restrict__ cand_th) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    int o = offsets[t];
    for (int h = 0; h < H; ++h) {
        if (!flags[(size_t)t * H + h]) continue;
        if (o < cap) {
            for (int j = 0; j < n; ++j) cand_q[(size_t)o * n + j] = trajT[((size_t)h * n + j) * N + t];
            cand_th[2 * o] = t;
            cand_th[2 * o + 1] = h;
        }
        ++o;
    }
}

template <int ND>
static void launch_cand_flags_t(hipStream_t s, int N, int H, int K, const float* trajT, const float* distT, const float* dotT,
                                const float* means, float a, float b, float c, float p, unsigned char* flags, int* counts) {
    hipLaunchKernelGGL(k_cand_flags<ND>, dim3((N + 255) / 256), dim3(256), 0, s, N, H, K, trajT, distT, dotT, means, a, b, c, p,
                       flags, counts);
}

void omds_launch_candidates(hipStream_t s, int N, int H, int n, int K, const float* trajT, const float* distT,
                            const float* dotT, const float* means, float thr_dist, float thr_kernel, float thr_dot,
                            float rbf_p, unsigned char* flags, int* counts, int* offsets, int cap, float* cand_q,
                            int* cand_th) {
    switch (n) {
        case 1: launch_cand_flags_t<1>(s, N, H, K, trajT, distT, dotT, means, thr_dist, thr_kernel, thr_dot, rbf_p, flags, counts); break;
        case 2: launch_cand_flags_t<2>(s, N, H, K, trajT, distT, dotT, means, thr_dist, thr_kernel, thr_dot, rbf_p, flags, counts); break;
        case 3: launch_cand_flags_t<3>(s, N, H, K, trajT, distT, dotT, means, thr_dist, thr_kernel, thr_dot, rbf_p, flags, counts); break;
        case 4: launch_cand_flags_t<4>(s, N, H, K, trajT, distT, dotT, means, thr_dist, thr_kernel, thr_dot, rbf_p, flags, counts); break;
        case 5: launch_cand_flags_t<5>(s, N, H, K, trajT, distT, dotT, means, thr_dist, thr_kernel, thr_dot, rbf_p, flags, counts); break;
        case 6: launch_cand_flags_t<6>(s, N, H, K, trajT, distT, dotT, means, thr_dist, thr_kernel, thr_dot, rbf_p, flags, counts); break;
        default: launch_cand_flags_t<7>(s, N, H, K, trajT, distT, dotT, means, thr_dist, thr_kernel, thr_dot, rbf_p, flags, counts); break;
    }
    hipLaunchKernelGGL(k_cand_scan, dim3(1), dim3(1024), 0, s, counts, N, offsets);
    hipLaunchKernelGGL(k_cand_write, dim3((N + 255) / 256), dim3(256), 0, s, N, H, n, trajT, flags, offsets, cap, cand_q, cand_th);
}

// ------------------------------------------------------------------------------------------------
// layout conversions
// ------------------------------------------------------------------------------------------------
__global__ void k_broadcast_q(const float* __restrict__ q, int n, int N, float* __restrict__ dstT) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    for (int j = 0; j < n; ++j) dstT[(size_t)j * N + t] = q[j];
}
void omds_launch_broadcast_q(hipStream_t s, const float* q_dev, int n, int N, float* dstT) {
    hipLaunchKernelGGL(k_broadcast_q, dim3((N + 255) / 256), dim3(256), 0, s, q_dev, n, N, dstT);
}

// dst[c][r] = src[r][c]; 32x32 LDS tiles
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int yy = ty; yy < 32; yy += 8) {
        const int r = by + yy, c = bx + tx;
        tile[yy][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int yy = ty; yy < 32; yy += 8) {
        const int c = bx + yy, r = by + tx;
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[tx][yy];
    }
}
void omds_launch_transpose(hipStream_t s, const float* src, float* dst, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(k_transpose, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, src, dst, rows, cols);
}

// dst[t][h][x] = srcT[(h*Xld + x)*N + t]  for x < X   (device SoA -> reference [N,H,X])
__global__ __launch_bounds__(256) void k_permute(const float* __restrict__ srcT, float* __restrict__ dst, int H, int X, int N,
                                                 int Xld) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)N * H * X;
    if (e >= total) return;
    const int x = (int)(e % X);
    const size_t th = e / X;
    const int h = (int)(th % H);
    const int t = (int)(th / H);
    dst[e] = srcT[((size_t)h * Xld + x) * N + t];
}
void omds_launch_permute_hxn_to_nhx(hipStream_t s, const float* srcT, float* dst, int H, int X, int N, int Xld) {
    const size_t total = (size_t)N * H * X;
    if (total == 0) return;
    hipLaunchKernelGGL(k_permute, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, srcT, dst, H, X, N, Xld);
}

// dst[r][h][x] = srcT[(h*Xld + x)*N + tlist[r]]  for x < X: the rows of a few rollouts in the reference layout (omds_get_rollout_rows)
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ srcT, float* __restrict__ dst, const int* __restrict__ tlist,
                                                     int count, int H, int X, int N, int Xld) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count * H * X) return;
    const int x = e % X, rh = e / X, h = rh % H, r = rh / H;
    dst[e] = srcT[((size_t)h * Xld + x) * N + tlist[r]];
}
void omds_launch_gather_rows(hipStream_t s, const float* srcT, float* dst, const int* tlist, int count, int H, int X, int N, int Xld) {
    const int total = count * H * X;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_gather_rows, dim3((total + 255) / 256), dim3(256), 0, s, srcT, dst, tlist, count, H, X, N, Xld);
}

// What the horizon loop accumulates for the update (step_device.h: maxact, phisum0), from caller-supplied tensors in the reference
// layouts (omds_weighted_update_eval): maxact[kk][t] = max_h(phi * act) with the loop's NaN rule, phisum0[kk] = sum_h phi of rollout 0
// in the loop's order.  kval_is_product: OMDS_VARIANT_KVAL_TIMES_ACT (the stored kernel values already hold phi * act).
__global__ __launch_bounds__(256) void k_update_inputs(const float* __restrict__ kval, const float* __restrict__ act, int N, int H, int K,
                                                       int kval_is_product, float* __restrict__ maxact, float* __restrict__ phisum0) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * K) return;
    const int t = e / K, kk = e % K;
    float m = 0.f, s = 0.f;
    for (int h = 0; h < H; ++h) {
        const float phi = kval[((size_t)t * H + h) * K + kk];
        const float pa = kval_is_product ? phi : phi * act[(size_t)t * H + h];
        m = (h == 0) ? pa : ((m != m || pa != pa) ? __builtin_nanf("") : fmaxf(m, pa));
        s = (h == 0 ? 0.f : s) + phi;
    }
    maxact[(size_t)kk * N + t] = m;
    if (t == 0) phisum0[kk] = s;
}
void omds_launch_update_inputs(hipStream_t s, const float* kval, const float* act, int N, int H, int K, int kval_is_product, float* maxact,
                               float* phisum0) {
    if (N * K <= 0) return;
    hipLaunchKernelGGL(k_update_inputs, dim3((N * K + 255) / 256), dim3(256), 0, s, kval, act, N, H, K, kval_is_product, maxact, phisum0);
}

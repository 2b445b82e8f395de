// Fused horizon step for scenes with FEW obstacles (O <= 32; planar 7-DoF: O = 8, k = 1), gfx950.
//
// With O obstacles a rollout has only O network rows, so a 32-row MFMA tile holds ALL rows of R = floor(32 / O) rollouts
// and the whole step of those rollouts -- MPPI.distance_repulsion_nn (MPPI.py:227-282) + the modulation / policy / Euler
// step (MPPI.py:102-223) -- needs no other workgroup: ONE launch per horizon step, one workgroup per R rollouts, no
// min-distance matrix, no candidate list, no second forward:
//
//   1. pass1_tile (MODE 2): the fp32 forward of the R*O pairs on v_mfma_f32_32x32x2 -- the arithmetic of k_pass1, bit for
//      bit -- leaving per row the pass-1 value D (min over the un-ignored links), the pass-2 distance and arg-min link
//      (over all outputs, robot_sdf.py:155) and the ReLU masks of every layer in LDS.
//   2. per rollout: the k smallest D by (D, obstacle) -- its O values sit in one wave.
//   3. the backward on the R*k <= 4 selected rows.  A 4-row backward is no MFMA problem: a 16-row tile would run at the
//      16-row rate (13.6 us for the four hidden layers on one CU) to move four useful rows.  It is a weight-streaming
//      problem -- 256 KB per layer through one CU at 64 B/clk = 1.7 us -- so it runs on the VALU: thread (jq, kp) owns output
//      columns 4jq..4jq+3 and the k range 32kp..32kp+31, reads W[k][4jq..] as one 16-byte load (1 KiB per wave instruction,
//      row-major weights as torch stores them), the four rows' gradients at k as one broadcast ds_read_b128, 16 FMAs; the
//      eight k parts meet in LDS.  Masks come from step 1's ballots.
//   4. first-layer backward, positional-encoding chain rule, then blend / modulation / Euler step / outputs / next-step
//      layer 1 exactly as k_tail does (modulate_core).
//
// Used when the scene qualifies (ReLU network without skips, O <= 32, R*k <= 4) and the batch is small enough that the
// two-kernel step would leave its tail on a fraction of the CUs (omds_step_small_wanted).
#include <algorithm>

#include "mlp_device.h"
#include "step_device.h"

constexpr int SS_RK = 4;      // backward rows per workgroup (R * k)
constexpr int SS_NT = 512;

struct SmallArgs {
    MlpDev m;
    const float* Bpre;
    const float* radius;
    const float* xyzr;
    float* Apre;          // [N][256] in: this step, out: next step
    int O, R;             // obstacles, rollouts per workgroup (R * O <= 32, R * k <= SS_RK)
    uint32_t ignored;
    OmdsDivisor odiv;
    int B;                // rollouts (states) of this launch
    const float* qT;      // [n][ldq] their joint states, transposed
    int ldq;
    // network-only form (omds_dist_grad; the same arithmetic as the step, so the batch entry point reproduces the step bit
    // for bit): gradients / distances / indices of the k selected rows and the pass-1 matrix go to global memory and the
    // kernel returns before the modulation.  o_gradx == nullptr: the step form.
    float* o_gradx;       // [B*k][d]
    float* o_drow;        // [B*k]
    int32_t* o_idx;       // [B][k]
    float* o_Dmin;        // [B][O]
    int dbg_stop;         // timing experiments only (OMDS_SMALL_STOP): return after phase 1 / 2 / 3 / 4
    StepArgs st;
};

__device__ __forceinline__ uint32_t ss_mask_bit(const uint32_t* maskS, int nhid, int row, int level, int col) {
    const uint32_t* mr = maskS + ((size_t)row * nhid + level) * 8;
    // level 0 (layer 1): ballot of component col & 3 of lane col >> 2; levels >= 1: one ballot half per 32-column block
    return level == 0 ? (mr[(col & 3) * 2 + (col >> 7)] >> ((col >> 2) & 31)) & 1u : (mr[col >> 5] >> (col & 31)) & 1u;
}

template <int ND, int TR>
__global__ __launch_bounds__(SS_NT, TR == 16 ? 4 : 2) void k_step_small(SmallArgs a) {
    constexpr int ACT = OMDS_ACT_RELU;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const MlpDev& m = a.m;
    const int nhid = m.nhh + 1;
    // pass1_tile's block: Hs [TR][LDH], rowRad [TR], rowIdx [TR], maskS [TR][nhid][8]
    float* Hs = smem;
    uint32_t* maskS = reinterpret_cast<uint32_t*>(smem + TR * LDH + 2 * TR);
    float* D1 = reinterpret_cast<float*>(maskS + TR * nhid * 8);     // [32] pass-1 value of each tile row
    float* Dr = D1 + 32;                                             // [32] pass-2 distance
    int* Amin = reinterpret_cast<int*>(Dr + 32);                     // [32] arg-min link
    int* selRow = Amin + 32;                                         // [SS_RK] tile row of each backward row (-1: none)
    int* selT = selRow + SS_RK;                                      // [SS_RK] rollout
    int* selO = selT + SS_RK;                                        // [SS_RK] obstacle
    float* dr = reinterpret_cast<float*>(selO + SS_RK);              // [SS_RK] pass-2 distance of each backward row
    float* gx = dr + SS_RK;                                          // [SS_RK][12] input gradients
    float* gf = gx + SS_RK * 12;                                     // [SS_RK][33] feature gradients
    float* feat = gf + SS_RK * 33;                                   // [SS_RK][3 ND] next state, sin, cos
    float4* gS = reinterpret_cast<float4*>(feat + SS_RK * 3 * OMDS_MAX_DOF + 4);   // [256] gradient of the four rows at each column
    // [8 k parts][256] partial sums: in the 32-row tile buffer (idle by then); the 16-row buffer is too small for them
    float4* P = TR == 32 ? reinterpret_cast<float4*>(Hs) : gS + OMDS_WIDTH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.B, k = a.st.k, O = a.O, R = a.R;
    const int t_base = blockIdx.x * R;

    // ---- 1. forward of the R*O pairs -----------------------------------------------------------------------------
    {
        const ExactOut ex{D1, Dr, Amin, nullptr, TR};
        const long long total = (long long)N * O;
        pass1_tile<TR, 1, 1, ACT, 2>(m, smem, a.Apre, a.Bpre, a.radius, O, total, a.ignored, nullptr, (long long)t_base * O, a.odiv,
                                     nullptr, nullptr, &ex);
    }
    if (a.dbg_stop == 1) return;
    if (tid < SS_RK) { selRow[tid] = -1; selT[tid] = 0; selO[tid] = 0; dr[tid] = 0.f; }
    __syncthreads();

    // ---- 2. the k closest obstacles of each rollout by (D, obstacle index) ------------------------------------------
    for (int rl = wave; rl < R; rl += 8) {
        const int t = t_base + rl;
        if (t >= N) break;
        const float x = lane < O ? D1[rl * O + lane] : __builtin_inff();
        float pv = -__builtin_inff();
        int po = -1;
        for (int j = 0; j < k; ++j) {
            const bool after = (x > pv) || (x == pv && lane > po);
            float bv = (lane < O && after) ? x : __builtin_inff();
            int bo = (lane < O && after) ? lane : 0x7fffffff;
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {   // O <= 32: the values sit in lanes 0-31
                const float ov = __shfl_xor(bv, off);
                const int oo = __shfl_xor(bo, off);
                if ((ov < bv) || (ov == bv && oo < bo)) { bv = ov; bo = oo; }
            }
            pv = bv;
            po = bo;
            if (lane == 0 && bo < O) {
                const int r = rl * k + j;
                selRow[r] = rl * O + bo;
                selT[r] = t;
                selO[r] = bo;
                dr[r] = Dr[rl * O + bo];
            }
        }
    }
    __syncthreads();

    if (a.dbg_stop == 2) return;
    // ---- 3. backward on the selected rows (VALU, weights streamed once) -------------------------------------------------
    // 16 weight rows per thread in flight at once (two halves of its k range: 117 registers instead of 151, the same time);
    // packed FMAs, two rows per instruction.  Measured: 4.2 us per layer whether or not the next layer's rows are requested
    // ahead of the reduction, and the same with scalar FMAs -- 256 workgroups pulling the same 256 KB through their L1s at
    // once is an L2 problem (32 CUs per XCD on the same lines).
    // Measured and rejected: a RESIDENT form of this kernel (one launch per propagate, the workgroup keeps its rollouts, their
    // layer-1 halves in LDS and their navigation kernels in registers over all H steps): 1.82 ms per iteration against 1.74 ms
    // for H launches -- consecutive launches already start with no idle gap, and the loop costs registers (245).
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int jq = tid & 63, kp = tid >> 6;
    float4 w[16];
    auto load_w = [&](int l, int half) {   // rows 32 kp + 16 half .. + 15 of the thread's k range (16 x 16 B per thread in flight)
        const float4* Wq = reinterpret_cast<const float4*>(m.Whraw + (size_t)l * OMDS_WIDTH * OMDS_WIDTH) + jq;   // W[k][4jq..4jq+3] = Wq[k * 64]
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = Wq[(size_t)(32 * kp + 16 * half + u) * 64];
    };
    if (tid < OMDS_WIDTH) {   // seed: dy[argmin] / dH_last = Wlast[argmin], masked by the last hidden layer
        float v[SS_RK];
#pragma unroll
        for (int r = 0; r < SS_RK; ++r) {
            const int row = selRow[r];
            v[r] = (row >= 0 && ss_mask_bit(maskS, nhid, row, m.nhh, tid)) ? m.Wlraw[(size_t)Amin[row] * OMDS_WIDTH + tid] : 0.f;
        }
        gS[tid] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
#pragma unroll 1
    for (int l = m.nhh - 1; l >= 0; --l) {
        f2 acc[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc[c][0] = f2{0.f, 0.f}; acc[c][1] = f2{0.f, 0.f}; }
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            load_w(l, half);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float4 g = gS[32 * kp + 16 * half + u];
                const f2 glo = {g.x, g.y}, ghi = {g.z, g.w};
                const float wc[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f2 ww = {wc[c], wc[c]};
                    acc[c][0] = __builtin_elementwise_fma(glo, ww, acc[c][0]);
                    acc[c][1] = __builtin_elementwise_fma(ghi, ww, acc[c][1]);
                }
                if ((u & 3) == 3) asm volatile("" ::: "memory");   // keeps hipcc from reading all gradient rows ahead of the FMAs
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) P[kp * OMDS_WIDTH + 4 * jq + c] = make_float4(acc[c][0][0], acc[c][0][1], acc[c][1][0], acc[c][1][1]);
        __syncthreads();
        if (tid < OMDS_WIDTH) {
            float4 s = P[tid];
#pragma unroll
            for (int q = 1; q < 8; ++q) {
                const float4 p = P[q * OMDS_WIDTH + tid];
                s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
            }
            const float sv[4] = {s.x, s.y, s.z, s.w};
            float v[SS_RK];
#pragma unroll
            for (int r = 0; r < SS_RK; ++r) {
                const int row = selRow[r];
                v[r] = (row >= 0 && ss_mask_bit(maskS, nhid, row, l, tid)) ? sv[r] : 0.f;
            }
            gS[tid] = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
    }
    if (a.dbg_stop == 3) return;
    // first layer: g_f[r][f] = sum_c Gz1[r][c] W1[c][f]; 16 lanes per feature, c strided over them
    {
        const int f = tid >> 4, sub = tid & 15, F = 3 * m.d;
        float s[SS_RK] = {};
        if (f < F) {
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int c = sub + 16 * i;
                const float w = m.W1t[(size_t)f * OMDS_WIDTH + c];
                const float4 g = gS[c];
                s[0] = fmaf(g.x, w, s[0]); s[1] = fmaf(g.y, w, s[1]); s[2] = fmaf(g.z, w, s[2]); s[3] = fmaf(g.w, w, s[3]);
            }
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1)
#pragma unroll
            for (int r = 0; r < SS_RK; ++r) s[r] += __shfl_xor(s[r], off);
        if (sub == 0 && f < 32) {
#pragma unroll
            for (int r = 0; r < SS_RK; ++r) gf[r * 33 + f] = (f < F) ? s[r] : 0.f;
        }
    }
    __syncthreads();
    const float* qT = a.qT;
    const int ldq = a.ldq;
    {   // positional-encoding chain rule: d/dx = g[x] + g[sin x] cos x - g[cos x] sin x
        const int d = m.d, n = m.n_dof;
        if (tid < SS_RK * d) {
            const int r = tid / d, jj = tid - r * d;
            if (selRow[r] >= 0) {
                const float x = (jj < n) ? qT[(size_t)jj * ldq + selT[r]] : a.xyzr[selO[r] * 4 + (jj - n)];
                gx[r * d + jj] = gf[r * 33 + jj] + gf[r * 33 + d + jj] * cosf(x) - gf[r * 33 + 2 * d + jj] * sinf(x);
            }
        }
    }
    __syncthreads();
    if (a.dbg_stop == 4) return;
    if (a.o_gradx != nullptr) {   // network-only form
        const int d = m.d;
        for (int e = tid; e < R * k * d; e += SS_NT) {
            const int r = e / d, jj = e - r * d;
            if (selRow[r] >= 0) a.o_gradx[((size_t)selT[r] * k + (r % k)) * d + jj] = gx[e];
        }
        if (tid < R * k && selRow[tid] >= 0) {
            a.o_drow[(size_t)selT[tid] * k + (tid % k)] = dr[tid];
            if (a.o_idx) a.o_idx[(size_t)selT[tid] * k + (tid % k)] = selO[tid];
        }
        if (a.o_Dmin)
            for (int e = tid; e < R * O; e += SS_NT)
                if (t_base + e / O < N) a.o_Dmin[(size_t)t_base * O + e] = D1[e];
        return;
    }

    // ---- 4. modulation / policy / Euler step: 16 lanes per rollout (as k_tail) -----------------------------------------
    {
        const int rl = tid >> 4, sub = tid & 15;
        const int t = t_base + rl;
        if (rl < R && t < N) {
            float q[ND], qn[ND];
#pragma unroll
            for (int j = 0; j < ND; ++j) q[j] = qT[(size_t)j * ldq + t];
            modulate_core<ND, 16>(a.st, a.st.step, t, sub, gx, dr, rl * k, q, qn);
            if (sub < ND) {
                float v = qn[0];
#pragma unroll
                for (int j = 1; j < ND; ++j) v = (sub == j) ? qn[j] : v;
                feat[rl * 3 * ND + sub] = v;
                feat[rl * 3 * ND + ND + sub] = sinf(v);
                feat[rl * 3 * ND + 2 * ND + sub] = cosf(v);
            }
        }
    }
    if (a.st.step >= a.st.H || a.dbg_stop == 5) return;   // last step: nothing is integrated, no next network evaluation
    __syncthreads();
    {   // rollout half of layer 1 for the next step (same arithmetic order as k_rollout_layer1)
        const int c = tid & 255, d = m.d;
        for (int rl = tid >> 8; rl < R; rl += 2) {
            const int t = t_base + rl;
            if (t >= N) break;
            const float* f = feat + rl * 3 * ND;
            float acc = m.b1[c];
#pragma unroll
            for (int j = 0; j < ND; ++j) acc = fmaf(m.W1t[(size_t)j * OMDS_WIDTH + c], f[j], acc);
#pragma unroll
            for (int j = 0; j < ND; ++j) acc = fmaf(m.W1t[(size_t)(d + j) * OMDS_WIDTH + c], f[ND + j], acc);
#pragma unroll
            for (int j = 0; j < ND; ++j) acc = fmaf(m.W1t[(size_t)(2 * d + j) * OMDS_WIDTH + c], f[2 * ND + j], acc);
            a.Apre[(size_t)t * OMDS_WIDTH + c] = acc;
        }
    }
}

static size_t small_lds_bytes(int nhid, int TR) {
    return ((size_t)TR * LDH + 2 * TR + (size_t)TR * nhid * 8) * 4 + (32 * 3 + SS_RK * 3) * 4 +
           (SS_RK + SS_RK * 12 + SS_RK * 33 + SS_RK * 3 * OMDS_MAX_DOF + 4) * 4 + OMDS_WIDTH * 16 + 16 + (TR == 16 ? 8 * OMDS_WIDTH * 16 : 0);
}

// rollouts per workgroup for (O, k) on a tile of `rows` rows, 0 = the scene does not qualify
static int small_rollouts(const MlpDev& m, int n_dof, int O, int k, int rows) {
    if (m.act != OMDS_ACT_RELU || m.skip_mask || (n_dof != 7 && n_dof != 2)) return 0;
    if (O < 1 || O > rows || k < 1 || k > SS_RK || k > O) return 0;
    return std::max(1, std::min(rows / O, SS_RK / k));
}
int omds_step_small_rollouts(const MlpDev& m, int n_dof, int O, int k) { return small_rollouts(m, n_dof, O, k, 32); }
// Tile height: 16 rows when that costs no rollouts per workgroup (the cap of four backward rows binds, not the tile: half the
// forward's MFMA chain for the same work); else 32.  Halving the rollouts per workgroup to get two 16-row workgroups resident
// per CU (one's backward and modulation under the other's forward) was measured on planar 7-DoF 1024 x 32: 17.0 M against
// 18.9 M rollout-steps/s -- twice the workgroups stream the backward's weights twice.  OMDS_SMALL_ROWS=16|32 forces one.
static int small_tile_rows(const MlpDev& m, int n_dof, int O, int k, int B) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("OMDS_SMALL_ROWS"); forced = e ? atoi(e) : 0; }
    const int r16 = small_rollouts(m, n_dof, O, k, 16), r32 = small_rollouts(m, n_dof, O, k, 32);
    if (r16 <= 0) return 32;
    if (forced == 16 || forced == 32) return forced;
    (void)B;
    return r16 == r32 ? 16 : 32;
}

template <int ND, int TR>
static void launch_small_r(hipStream_t s, const SmallArgs& a) {
    static std::atomic<uint64_t> configured{0};
    if (omds_first_use_on_device(configured))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_small<ND, TR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)small_lds_bytes(OMDS_MAX_HIDDEN + 1, TR));
    hipLaunchKernelGGL((k_step_small<ND, TR>), dim3((a.B + a.R - 1) / a.R), dim3(SS_NT), small_lds_bytes(a.m.nhh + 1, TR), s, a);
}
template <int ND>
static void launch_small_t(hipStream_t s, SmallArgs& a, int k) {
    const int TR = small_tile_rows(a.m, ND, a.O, k, a.B);
    a.R = small_rollouts(a.m, ND, a.O, k, TR);
    if (a.R <= 0) return;
    if (TR == 16) launch_small_r<ND, 16>(s, a);
    else launch_small_r<ND, 32>(s, a);
}

void omds_launch_step_small(hipStream_t s, const MlpDev& m, const float* Bpre, const float* radius, const float* xyzr, float* Apre,
                            int O, uint32_t ignored, const StepArgs& st) {
    SmallArgs a{};
    a.m = m; a.Bpre = Bpre; a.radius = radius; a.xyzr = xyzr; a.Apre = Apre; a.O = O; a.ignored = ignored;
    a.R = omds_step_small_rollouts(m, st.n, O, st.k);
    a.odiv = OmdsDivisor::make((unsigned)O);
    a.B = st.N;
    a.qT = st.trajT + (size_t)(st.step - 1) * st.n * st.N;
    a.ldq = st.N;
    a.st = st;
    static int stop = -1;
    if (stop < 0) { const char* e = getenv("OMDS_SMALL_STOP"); stop = e ? atoi(e) : 0; }
    a.dbg_stop = stop;
    if (omds_step_small_rollouts(m, st.n, O, st.k) <= 0) return;
    if (st.n == 7) launch_small_t<7>(s, a, st.k);
    else launch_small_t<2>(s, a, st.k);
}

// the network part alone on B states (qT [n][ldq]); Apre holds their layer-1 halves
void omds_launch_net_small(hipStream_t s, const MlpDev& m, const float* Bpre, const float* radius, const float* xyzr, float* Apre,
                           int O, uint32_t ignored, int n_dof, int k, const float* qT, int ldq, int B, float* gradx, float* drow,
                           int32_t* idx, float* Dmin) {
    SmallArgs a{};
    a.m = m; a.Bpre = Bpre; a.radius = radius; a.xyzr = xyzr; a.Apre = Apre; a.O = O; a.ignored = ignored;
    a.R = omds_step_small_rollouts(m, n_dof, O, k);
    a.odiv = OmdsDivisor::make((unsigned)O);
    a.B = B; a.qT = qT; a.ldq = ldq;
    a.o_gradx = gradx; a.o_drow = drow; a.o_idx = idx; a.o_Dmin = Dmin;
    a.st.k = k; a.st.n = n_dof; a.st.N = B; a.st.d = m.d;
    if (omds_step_small_rollouts(m, n_dof, O, k) <= 0 || B <= 0) return;
    if (n_dof == 7) launch_small_t<7>(s, a, k);
    else launch_small_t<2>(s, a, k);
}

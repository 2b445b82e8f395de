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
    const float* Fp;
    const float* radius;
    const float* xyzr;
    float* Fq;          // [N][OMDS_FROW] encoded joint inputs, in: this step, out: next step
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
    const uint32_t* mr = maskS + ((size_t)row * nhid + level) * 8;   // one ballot half per 32-column block
    return (mr[col >> 5] >> (col & 31)) & 1u;
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
    float* gf = gx + SS_RK * 12;                                     // [16][33] feature gradients (first_layer_backward works on a 16-row block)
    float* feat = gf + 16 * 33;                                      // [SS_RK][3 ND] next state, sin, cos
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.B, k = a.st.k, O = a.O, R = a.R;
    const int t_base = blockIdx.x * R;

    // ---- 1. forward of the R*O pairs -----------------------------------------------------------------------------
    {
        const ExactOut ex{D1, Dr, Amin, nullptr, TR};
        const long long total = (long long)N * O;
        pass1_tile<TR, 1, 1, ACT, 2>(m, smem, a.Fq, a.Fp, a.radius, O, total, a.ignored, nullptr, (long long)t_base * O, a.odiv,
                                     nullptr, nullptr, &ex);
    }
    if (OMDS_DBG(a.dbg_stop) == 1) return;
    if (tid < SS_RK) { selRow[tid] = -1; selT[tid] = 0; selO[tid] = 0; dr[tid] = 0.f; }
    // the backward's first weights (phase 3) are on their way while the closest obstacles are picked
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const bool mine = wv < 4;    // the GEMM waves, one per SIMD
    W4Ring<8> ring;
    if (mine) {
        ring.bind(m.Wb4, m.nhh, wv, lane);
        ring.fill(m.nhh - 1);
    }
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

    if (OMDS_DBG(a.dbg_stop) == 2) return;
    // ---- 3. backward on the selected rows: ONE 4-row group on v_mfma_f32_4x4x1 (gemm4, mlp_device.h) -----------------------
    // Four rows are no 16-row MFMA problem (a 16-row tile would run at the 16-row rate to move four useful rows) but they are
    // exactly one row group of the 4x4x1 shape: waves 0-3 multiply 64 columns each, 256 dependent MFMAs per layer (one
    // accumulator chain in the k order of every other kernel: the gradient rows are the two-launch step's bit for bit), the
    // 256 KB of a layer streaming through a ring of 8 chunks that runs across the layers (1.9 us per layer through the CU's
    // L1 -- the bound).  The first version ran on the VALU with the k range split over the eight waves: 16 rows per thread
    // loaded, waited for, 128 packed FMAs, then eight partial sums meeting in LDS -- 7.2 k cycles per layer of which 3.4 k were
    // the partial-sum write, two barriers and the reduction.
    // Measured and rejected: the modulation's gradient-independent half (nominal DS, goal activation, RBF policy) on wave 0 under
    // the first GEMM, the GEMM on waves 4-7 -- the GEMM of the SIMD that wave 0 shares grew by the whole 2.7 us (19.9 -> 19.1 M):
    // even a single dependent 4x4x1 chain does not share its SIMD's issue with another wave's VALU / transcendental stream.
    // Measured and rejected: a RESIDENT form of this kernel (one launch per propagate, the workgroup keeps its rollouts, their
    // layer-1 halves in LDS and their navigation kernels in registers over all H steps): 1.82 ms per iteration against 1.74 ms
    // for H launches -- consecutive launches already start with no idle gap, and the loop costs registers (245).
    {
        const int col = 64 * (wv & 3) + lane, pcol = omds_kpos(col);   // the column and its position in the k-permuted tile
        // this thread's ReLU masks, all levels: bit 4 l + r = row r at level l (the GEMM waves: their column)
        uint32_t mb = 0;
        int amin[SS_RK];
        if (mine) {
#pragma unroll
            for (int r = 0; r < SS_RK; ++r) {
                const int row = selRow[r];
                amin[r] = row >= 0 ? Amin[row] : 0;
                for (int l = 0; l < nhid; ++l)
                    if (row >= 0 && ss_mask_bit(maskS, nhid, row, l, col)) mb |= 1u << (4 * l + r);
            }
        }
        if (mine) {        // seed: dy[argmin] / dH_last = Wlast[argmin], masked by the last hidden layer (the tile buffer is idle since the forward)
#pragma unroll
            for (int r = 0; r < SS_RK; ++r)
                Hs[r * LDH + pcol] = ((mb >> (4 * m.nhh + r)) & 1u) ? m.Wlraw[(size_t)amin[r] * OMDS_WIDTH + col] : 0.f;
        }
        __syncthreads();
#pragma unroll 1
        for (int l = m.nhh - 1; l >= 0; --l) {
            f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
            if (mine) {
                gemm4<1, 8>(Hs, ring, l, l - 1, lane, acc);
            }
            __syncthreads();   // every wave has read the rows
            if (mine) {
                float v[SS_RK];
#pragma unroll
                for (int r = 0; r < SS_RK; ++r) {
                    v[r] = ((mb >> (4 * l + r)) & 1u) ? acc[0][r] : 0.f;
                    Hs[r * LDH + pcol] = v[r];
                }
            }
            __syncthreads();
        }
    }
    if (OMDS_DBG(a.dbg_stop) == 3) return;
    // first layer: g_f[r][f] = sum_c Gz1[r][c] W1[c][f], one ascending chain per element over the four gradient rows at the top of
    // the tile (first_layer_backward works on 16-row blocks: rows 4..15 hold the forward's leftovers, whose products stay in their
    // own rows of the MFMA and are not read)
    first_layer_backward<16>(m, Hs, gf, nullptr);
    __syncthreads();
    const float* qT = a.qT;
    const int ldq = a.ldq;
    {   // positional-encoding chain rule: d/dx = g[x] + g[sin x] cos x - g[cos x] sin x
        const int d = m.d, n = m.n_dof;
        if (tid < SS_RK * d) {
            const int r = tid / d, jj = tid - r * d;
            if (selRow[r] >= 0) {
                const float x = (jj < n) ? qT[(size_t)jj * ldq + selT[r]] : a.xyzr[selO[r] * 4 + (jj - n)];
                gx[r * d + jj] = pe_chain_rule(gf[r * 33 + jj], gf[r * 33 + d + jj], gf[r * 33 + 2 * d + jj], x);
            }
        }
    }
    __syncthreads();
    if (OMDS_DBG(a.dbg_stop) == 4) return;
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
                feat[rl * 3 * ND + ND + sub] = omds_sinf(v);
                feat[rl * 3 * ND + 2 * ND + sub] = omds_cosf(v);
            }
        }
    }
    if (a.st.step >= a.st.H || OMDS_DBG(a.dbg_stop) == 5) return;   // last step: nothing is integrated, no next network evaluation
    __syncthreads();
    {   // the encoded joint inputs of the next step (as k_rollout_features writes them)
        const int d = m.d;
        for (int e = tid; e < R * 3 * ND; e += SS_NT) {
            const int rl = e / (3 * ND), cc = e - rl * (3 * ND), part = cc / ND, t = t_base + rl;
            if (t < N) a.Fq[(size_t)t * OMDS_FROW + part * d + (cc - part * ND)] = feat[e];
        }
    }
}

static size_t small_lds_bytes(int nhid, int TR) {
    // the last term: gemm4's A-operand reads span 32 tile rows (lanes 16-31 are never selected, but they read): a 16-row tile
    // buffer is followed by at least another 16 rows' worth of allocation
    return ((size_t)TR * LDH + 2 * TR + (size_t)TR * nhid * 8) * 4 + (32 * 3 + SS_RK * 3) * 4 +
           (SS_RK + SS_RK * 12 + 16 * 33 + SS_RK * 3 * OMDS_MAX_DOF + 4) * 4 + 16 + (TR == 16 ? (size_t)16 * LDH * 4 : 0);
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
    static const int forced = OMDS_EXP_ENV("OMDS_SMALL_ROWS", 0);   // experiment builds
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

void omds_launch_step_small(hipStream_t s, const MlpDev& m, const float* Fp, const float* radius, const float* xyzr, float* Fq,
                            int O, uint32_t ignored, const StepArgs& st) {
    SmallArgs a{};
    a.m = m; a.Fp = Fp; a.radius = radius; a.xyzr = xyzr; a.Fq = Fq; a.O = O; a.ignored = ignored;
    a.R = omds_step_small_rollouts(m, st.n, O, st.k);
    a.odiv = OmdsDivisor::make((unsigned)O);
    a.B = st.N;
    a.qT = st.trajT + (size_t)(st.step - 1) * st.n * st.N;
    a.ldq = st.N;
    a.st = st;
    static const int stop = OMDS_EXP_ENV("OMDS_SMALL_STOP", 0);
    a.dbg_stop = stop;
    if (omds_step_small_rollouts(m, st.n, O, st.k) <= 0) return;
    if (st.n == 7) launch_small_t<7>(s, a, st.k);
    else launch_small_t<2>(s, a, st.k);
}

// the network part alone on B states (qT [n][ldq]); Fq holds their layer-1 halves
void omds_launch_net_small(hipStream_t s, const MlpDev& m, const float* Fp, const float* radius, const float* xyzr, float* Fq,
                           int O, uint32_t ignored, int n_dof, int k, const float* qT, int ldq, int B, float* gradx, float* drow,
                           int32_t* idx, float* Dmin) {
    SmallArgs a{};
    a.m = m; a.Fp = Fp; a.radius = radius; a.xyzr = xyzr; a.Fq = Fq; a.O = O; a.ignored = ignored;
    a.R = omds_step_small_rollouts(m, n_dof, O, k);
    a.odiv = OmdsDivisor::make((unsigned)O);
    a.B = B; a.qT = qT; a.ldq = ldq;
    a.o_gradx = gradx; a.o_drow = drow; a.o_idx = idx; a.o_Dmin = Dmin;
    a.st.k = k; a.st.n = n_dof; a.st.N = B; a.st.d = m.d;
    if (omds_step_small_rollouts(m, n_dof, O, k) <= 0 || B <= 0) return;
    if (n_dof == 7) launch_small_t<7>(s, a, k);
    else launch_small_t<2>(s, a, k);
}

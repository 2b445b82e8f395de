// Fused per-step "tail" kernel for gfx950: everything of one horizon step that is local to a rollout
// once the pass-1 min-distance matrix exists -- top-k over the obstacles (MPPI.py:245-247), forward +
// analytic backward on the k closest rows (robot_sdf.py:153-158), softmax blend, modulation / policy /
// Euler step (MPPI.py:102-223) and the encoded joint inputs [q, sin q, cos q] of the NEXT step.  A workgroup owns
// floor(ROWS/k) rollouts (<= ROWS network rows), so a horizon step is two launches: k_pass1 + k_tail.  ROWS = 32
// (v_mfma_f32_32x32x2) for large batches; ROWS = 16 (v_mfma_f32_16x16x4) when the batch has too few rows to fill the
// CUs with 32-row tiles anyway: the workgroup's chain of dependent GEMMs IS the step latency then, and it halves.
// The stand-alone kernels (k_topk, k_pass2, k_modulate, k_rollout_layer1) remain for the batch entry
// points (omds_dist_grad, omds_mlp_forward_vjp) and for n_dof / k combinations not instantiated here.
#include <algorithm>

#include <atomic>

#include "mlp_device.h"
#include "step_device.h"

struct TailArgs {
    MlpDev m;
    const float* Fp;
    const float* radius;
    const float* xyzr;
    const float* Dmin;   // [N][O]
    float* Fq;         // [N][OMDS_FROW] encoded joint inputs, in: this step, out: next step (rows of the workgroup's own rollouts)
    float* FqOut;      // where the next step's rows go: Fq itself, or the next slab when all steps' halves are kept
    _Float16* FqH;       // next step's states as fp16 network inputs for the screening kernel (nullptr: not screening)
    int ldF;             // row capacity of FqH
    float* dscr;         // tanh derivative scratch
    int O;
    int t_begin, t_end;  // rollout range of this launch (a group of the rollouts; the whole batch = [0, N))
    int slot0;           // first 32-row slot of this launch in the tanh scratch
    int n_slots;         // slots of the whole batch (scratch stride between layers)
    int dbg_stop;        // timing experiments only (OMDS_TAIL_STOP): return after phase 1 / 2 / 3
    // screened step (k_tail_sel): the candidate list of k_select and what k_exact computed for its entries
    const int* rowlist;  // [entries] pair t*O + o
    const int* range;    // [N][4] start, length of each rollout's entries, tau of k_select (float bits)
    float e_bound;       // assumed bound on the screening error of the rows that were NOT re-evaluated
    unsigned* viol;      // count of rollouts whose slack tau - (exact k-th smallest) fell below e_bound
    ExactOut ex;
    StepArgs st;
};

template <int ND, int ACT, int ROWS>
__global__ __launch_bounds__(P2_NT) void k_tail(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const MlpDev& m = a.m;
    P2Smem sm;
    sm.Hs = smem;
    sm.gf = sm.Hs + P2_MT * LDH;
    sm.maskL = reinterpret_cast<uint16_t*>(sm.gf + 32 * 33);
    sm.rowT = reinterpret_cast<int*>(sm.maskL + ((m.nhh + 2) / 2 * 2) * P2_NT);
    sm.rowO = sm.rowT + P2_MT;
    sm.rowMin = sm.rowO + P2_MT;
    float* gx = reinterpret_cast<float*>(sm.rowMin + P2_MT);   // [32][d]
    float* dr = gx + 32 * 12;                                   // [32]
    float* feat = dr + 32;                                      // [32][3*ND]: q_next, sin, cos
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.st.N, k = a.st.k, O = a.O;
    // ROWS = 4: forward and backward on 4-row groups (pass2_body_g4): 4 G4_NG = 20 network rows per workgroup
    static_assert(ACT == OMDS_ACT_RELU || ROWS != 4, "the 4-row-group pass 2 works on ReLU masks");
    constexpr int G4_NG = 5, TROWS = ROWS == 4 ? 4 * G4_NG : ROWS;
    const int RW = TROWS / k;                 // rollouts per workgroup
    const int t_base = a.t_begin + blockIdx.x * RW;
    const int t_end = a.t_end;

    OMDS_TL_STAMP(0);
    OMDS_TL_STAMP(1);
    P2G4Pre pre;   // (ROWS = 4) everything of pass 2 that does not wait for the top-k is requested in front of it
    if constexpr (ROWS == 4)
        pass2_g4_prefetch(m, a.Fq, [&](int r) { const int rl = r / k; return (rl < RW && t_base + rl < t_end) ? t_base + rl : -1; }, pre);
    // ---- top-k of each rollout's min-distance row (ascending, ties by lower obstacle index) -------
    if (tid < P2_MT) { sm.rowT[tid] = -1; sm.rowO[tid] = 0; }
    __syncthreads();
    for (int rl = wave; rl < RW; rl += 8) {
        const int t = t_base + rl;
        if (t >= t_end) break;
        topk_row(a.Dmin + (size_t)t * O, O, k, lane, [&](int j, int bi) {
            sm.rowT[rl * k + j] = t;
            sm.rowO[rl * k + j] = bi;
            // screened step of a tanh network: the matrix holds exact values on the candidates and screening values (all above
            // tau) elsewhere; the k smallest are exact ones, and no unevaluated row can belong among them, as long as the k-th
            // stays e_bound below tau (the slack guard)
            if (a.range && j == k - 1 && !(__builtin_bit_cast(float, a.range[4 * t + 2]) - a.Dmin[(size_t)t * O + bi] >= a.e_bound)) atomicAdd(a.viol, 1u);
        });
    }
    __syncthreads();
    OMDS_TL_STAMP(2);
    if (OMDS_DBG(a.dbg_stop) == 1) return;

    // ---- forward + backward on the selected rows; gradients and distances stay in LDS ------------
    const float* qT = a.st.trajT + (size_t)(a.st.step - 1) * ND * N;
    if constexpr (ROWS == 4)
        pass2_body_g4<G4_NG>(m, sm, pre, a.Fp, a.radius, a.xyzr, t_base * k, N * k, qT, N, gx, dr, 0, OMDS_DBG(a.dbg_stop));
    else
        pass2_body<ACT, ROWS>(m, sm, a.Fq, a.Fp, a.radius, a.xyzr, t_base * k, N * k, qT, N, gx, dr, 0, nullptr, nullptr,
                              a.dscr, (size_t)a.n_slots * ROWS * OMDS_WIDTH, (a.slot0 + (int)blockIdx.x) * ROWS, OMDS_DBG(a.dbg_stop));
    __syncthreads();
    OMDS_TL_STAMP(9);
    if (OMDS_DBG(a.dbg_stop) == 2) return;

    // ---- modulation / policy / Euler step: 16 lanes per rollout -------------------------------------
    {
        const int rl = tid >> 4, sub = tid & 15;
        const int t = t_base + rl;
        if (rl < RW && t < t_end) {
            float q[ND], qn[ND];
#pragma unroll
            for (int j = 0; j < ND; ++j) q[j] = qT[(size_t)j * N + t];
            modulate_core<ND, 16>(a.st, a.st.step, t, sub, gx, dr, rl * k, q, qn);
            if (sub < ND) {
                float v = qn[0];
#pragma unroll
                for (int j = 1; j < ND; ++j) v = (sub == j) ? qn[j] : v;
                feat[rl * 3 * ND + sub] = v;
                feat[rl * 3 * ND + ND + sub] = omds_sinf(v);
                feat[rl * 3 * ND + 2 * ND + sub] = omds_cosf(v);
                if (a.FqH && a.st.step < a.st.H) {
                    const int d = m.d;
                    a.FqH[omds_screen_fidx(sub, t, a.ldF)] = (_Float16)v;
                    a.FqH[omds_screen_fidx(d + sub, t, a.ldF)] = (_Float16)feat[rl * 3 * ND + ND + sub];
                    a.FqH[omds_screen_fidx(2 * d + sub, t, a.ldF)] = (_Float16)feat[rl * 3 * ND + 2 * ND + sub];
                    if (m.scrQ) {   // skip-connection networks: the concatenation operand of the screening kernel
                        _Float16* S = reinterpret_cast<_Float16*>(m.scrQ);
                        S[omds_screen_sidx(sub, 3 * d, t, a.ldF)] = (_Float16)v;
                        S[omds_screen_sidx(d + sub, 3 * d, t, a.ldF)] = (_Float16)feat[rl * 3 * ND + ND + sub];
                        S[omds_screen_sidx(2 * d + sub, 3 * d, t, a.ldF)] = (_Float16)feat[rl * 3 * ND + 2 * ND + sub];
                    }
                }
            }
        }
    }
    OMDS_TL_STAMP(10);
    if (a.st.step >= a.st.H || OMDS_DBG(a.dbg_stop) == 3) return;   // last step: nothing is integrated, no next network evaluation
    __syncthreads();

    // ---- the encoded joint inputs of the next step (as k_rollout_features writes them) --------
    {
        const int d = m.d;
        for (int e = tid; e < RW * 3 * ND; e += P2_NT) {
            const int rl = e / (3 * ND), cc = e - rl * (3 * ND), part = cc / ND, t = t_base + rl;
            if (t < t_end) a.FqOut[(size_t)t * OMDS_FROW + part * d + (cc - part * ND)] = feat[e];
        }
    }
    OMDS_TL_STAMP(11);
    OMDS_TL_STAMP(19);
}


static size_t tail_lds_bytes(int nhid);
#ifdef OMDS_TAIL_TL
__global__ void k_tail_tl_dump(int nb);
#endif

// ------------------------------------------------------------------------------------------------
// Screened step: k_screen -> k_select -> k_exact -> k_tail_sel.  k_exact has evaluated every candidate row in fp32 with
// pass 1's arithmetic, which is also pass 2's forward (same start value, same k order), and left the pass-1 value D, the
// pass-2 distance, the arg-min link and the ReLU masks per list entry.  This tail therefore needs no forward at all: top-k
// by (D, obstacle index) over each rollout's few candidates, masks of the selected entries gathered into the MFMA C layout,
// the pass-2 backward, then blend / modulation / Euler step / next-step layer 1 exactly as in k_tail.
// tanh networks (ACT = OMDS_ACT_TANH, round 4): k_exact left 1 - h^2 of every hidden unit of every candidate instead of masks
// (ExactOut::deriv, 1 KB per entry and layer); the backward reads the selected entries' rows through sm.rowE -- the same
// numbers pass 2's own forward would have written to its scratch, so the tanh tail drops its forward too.
// ------------------------------------------------------------------------------------------------
template <int ND, int ROWS, int ACT = OMDS_ACT_RELU>
__global__ __launch_bounds__(P2_NT) void k_tail_sel(TailArgs a) {
    static_assert(ACT == OMDS_ACT_RELU || ROWS != 4, "the 4-row-group backward works on ReLU masks");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const MlpDev& m = a.m;
    P2Smem sm;
    sm.Hs = smem;
    sm.gf = sm.Hs + P2_MT * LDH;
    sm.maskL = reinterpret_cast<uint16_t*>(sm.gf + 32 * 33);
    sm.rowT = reinterpret_cast<int*>(sm.maskL + ((m.nhh + 2) / 2 * 2) * P2_NT);
    sm.rowO = sm.rowT + P2_MT;
    sm.rowMin = sm.rowO + P2_MT;
    float* gx = reinterpret_cast<float*>(sm.rowMin + P2_MT);   // [32][d]
    float* dr = gx + 32 * 12;                                   // [32]
    float* feat = dr + 32;                                      // [32][3*ND]: q_next, sin, cos
    int* sel = reinterpret_cast<int*>(feat + 32 * 3 * OMDS_MAX_DOF);   // [32] list entry of each backward row (-1: padding)
    uint32_t* maskRow = reinterpret_cast<uint32_t*>(sm.Hs);     // [32][nhid][8] gathered masks; the tile buffer is idle until the seed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.st.N, k = a.st.k, O = a.O, nhid = m.nhh + 1;
    // ROWS = 4: the backward runs on 4-row groups (pass2_backward_hidden_g4): G4_ROWS network rows per workgroup
    constexpr int G4_NG = 5, TROWS = ROWS == 4 ? 4 * G4_NG : ROWS;
    const int RW = TROWS / k;                 // rollouts per workgroup
    const int t_base = a.t_begin + blockIdx.x * RW;
    const int t_end = a.t_end;

    OMDS_TL_STAMP(0);
    OMDS_TL_STAMP(1);
    // ---- top-k of each rollout's candidates by (D, obstacle index): k rounds of "smallest after the previous one" ----
    if (tid < P2_MT) { sm.rowT[tid] = -1; sm.rowO[tid] = 0; sm.rowMin[tid] = 0; sel[tid] = -1; dr[tid] = 0.f; }
    __syncthreads();
    for (int rl = wave; rl < RW; rl += 8) {
        const int t = t_base + rl;
        if (t >= t_end) break;
        // rowlist == nullptr: EVERY pair of the rollout is an entry and the entry index is the pair index (pass1_tile mode 6: the
        // all-fp32 step); no window, no slack to check
        const bool dense = a.rowlist == nullptr;
        const int base = dense ? t * O : a.range[4 * t], cnt = dense ? O : a.range[4 * t + 1];
        const float tau = dense ? __builtin_inff() : __builtin_bit_cast(float, a.range[4 * t + 2]);
        if (lane == 0 && cnt < k && a.viol) atomicAdd(a.viol, 1u);
        if (cnt <= 64) {
            // the usual case, a few candidates: one per lane, and its rank under (D, obstacle) by comparing with every other
            // candidate's (value, obstacle) broadcast from its lane -- cnt scalar steps instead of k wave-wide reductions
            const int e = base + lane;
            const bool have = lane < cnt && e < a.ex.cap;   // list longer than k_exact's outputs: the host redoes the propagate in fp32
            const float x = have ? a.ex.D[e] : __builtin_inff();
            const int o = have ? (dense ? e : a.rowlist[e]) - t * O : 0x7fffffff;
            int rank = 0;
            for (int j = 0; j < cnt; ++j) {
                const float xj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), j));
                const int oj = __builtin_amdgcn_readlane(o, j);
                rank += ((xj < x) || (xj == x && oj < o)) ? 1 : 0;
            }
            // slack guard: every row that was not listed has a screening value above tau; with an error of at most e_bound its
            // exact value exceeds tau - e_bound, so it stays out of the k smallest as long as tau - D*_k >= e_bound
            if (have && rank == k - 1 && a.viol && !(tau - x >= a.e_bound)) atomicAdd(a.viol, 1u);   // (dense mode: tau = inf, viol = NULL)
            if (have && rank < k) {
                const int r = rl * k + rank;
                sm.rowT[r] = t;
                sm.rowO[r] = o;
                sm.rowMin[r] = a.ex.amin[e];
                dr[r] = a.ex.dr[e];
                sel[r] = e;
            }
            continue;
        }
        float pv = -__builtin_inff();
        int po = -1;
        for (int j = 0; j < k; ++j) {
            float bv = __builtin_inff();
            int bo = 0x7fffffff, be = -1;
            for (int i = lane; i < cnt; i += 64) {
                const int e = base + i;
                if (e >= a.ex.cap) break;   // list longer than k_exact's outputs: the host sees the count and redoes the propagate in fp32
                const float x = a.ex.D[e];
                const int o = (dense ? e : a.rowlist[e]) - t * O;
                const bool after = (x > pv) || (x == pv && o > po);
                if (after && ((x < bv) || (x == bv && o < bo))) { bv = x; bo = o; be = e; }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ov = __shfl_xor(bv, off);
                const int oo = __shfl_xor(bo, off);
                const int oe = __shfl_xor(be, off);
                if ((ov < bv) || (ov == bv && oo < bo)) { bv = ov; bo = oo; be = oe; }
            }
            pv = bv;
            po = bo;
            if (lane == 0 && be >= 0) {
                const int r = rl * k + j;
                sm.rowT[r] = t;
                sm.rowO[r] = bo;
                sm.rowMin[r] = a.ex.amin[be];
                dr[r] = a.ex.dr[be];
                sel[r] = be;
            }
        }
        if (lane == 0 && a.viol && !(tau - pv >= a.e_bound)) atomicAdd(a.viol, 1u);   // pv = the exact k-th smallest
    }
    __syncthreads();
    OMDS_TL_STAMP(2);
    if (OMDS_DBG(a.dbg_stop) == 1) return;   // OMDS_TAIL_SEL_STOP: timing experiments
    if constexpr (ACT == OMDS_ACT_TANH) {
        sm.rowE = sel;   // the backward gathers 1 - h^2 of row r from entry sel[r] of ExactOut::deriv (written by k_exact, mode 5)
    } else {
    // ---- the selected entries' masks -> LDS -> 16 bits per thread and layer in the MFMA C layout (what pass 2's forward
    //      would have left in sm.maskL): thread (wave = 32-column block, lane) owns column wave*32 + (lane&31) of rows crow(r, lane)
    for (int i = tid; i < TROWS * nhid * 8; i += P2_NT) {
        const int r = i / (nhid * 8);
        maskRow[i] = sel[r] >= 0 ? a.ex.mask[(size_t)sel[r] * nhid * 8 + (i - r * nhid * 8)] : 0u;
    }
    __syncthreads();
    if constexpr (ROWS == 4) {
        // the 4-row-group backward wants, per hidden level and COLUMN, the mask bits of all rows in one word
        sm.maskG4 = reinterpret_cast<uint32_t*>(sm.maskL);   // (nhh + 1) * 256 words <= the (nhh + 1) * 512 halfwords of maskL
        for (int i = tid; i < nhid * 256; i += P2_NT) {
            const int l = i >> 8, col = i & 255;
            uint32_t bits = 0;
#pragma unroll
            for (int e = 0; e < 4 * G4_NG; ++e) {
                const uint32_t* mr = maskRow + ((size_t)e * nhid + l) * 8;
                bits |= ((mr[col >> 5] >> (col & 31)) & 1u) << e;
            }
            sm.maskG4[i] = bits;
        }
    } else {
        using G = P2Geo<ROWS>;
        for (int l = 0; l < nhid; ++l) {
            uint32_t bits = 0;
#pragma unroll
            for (int r = 0; r < G::NV; ++r) {
                const int col = G::col(r, wave, lane);
                const uint32_t* mr = maskRow + ((size_t)G::row(r, lane) * nhid + l) * 8;
                bits |= ((mr[col >> 5] >> (col & 31)) & 1u) << r;   // one ballot half per 32-column block and level
            }
            sm.maskL[l * P2_NT + tid] = (uint16_t)bits;
        }
    }
    __syncthreads();
    }

    OMDS_TL_STAMP(3);
    if (OMDS_DBG(a.dbg_stop) == 2) return;
    // ---- backward on the selected rows; gradients stay in LDS -----------------------------------------
    const float* qT = a.st.trajT + (size_t)(a.st.step - 1) * ND * N;
    if constexpr (ROWS == 4) {
        pass2_backward_hidden_g4<G4_NG>(m, sm);
        p2_backward_first<32>(m, sm, a.xyzr, t_base * k, N * k, qT, N, gx, 0);
    } else {
        pass2_backward<ACT, ROWS>(m, sm, a.xyzr, t_base * k, N * k, qT, N, gx, 0, a.ex.deriv, (size_t)a.ex.cap * OMDS_WIDTH, 0);
    }
    __syncthreads();
    OMDS_TL_STAMP(9);
    if (OMDS_DBG(a.dbg_stop) == 3) return;

    // ---- modulation / policy / Euler step: 16 lanes per rollout -------------------------------------
    {
        const int rl = tid >> 4, sub = tid & 15;
        const int t = t_base + rl;
        if (rl < RW && t < t_end) {
            float q[ND], qn[ND];
#pragma unroll
            for (int j = 0; j < ND; ++j) q[j] = qT[(size_t)j * N + t];
            modulate_core<ND, 16>(a.st, a.st.step, t, sub, gx, dr, rl * k, q, qn);
            if (sub < ND) {
                float v = qn[0];
#pragma unroll
                for (int j = 1; j < ND; ++j) v = (sub == j) ? qn[j] : v;
                feat[rl * 3 * ND + sub] = v;
                feat[rl * 3 * ND + ND + sub] = omds_sinf(v);
                feat[rl * 3 * ND + 2 * ND + sub] = omds_cosf(v);
                if (a.FqH && a.st.step < a.st.H) {
                    const int d = m.d;
                    a.FqH[omds_screen_fidx(sub, t, a.ldF)] = (_Float16)v;
                    a.FqH[omds_screen_fidx(d + sub, t, a.ldF)] = (_Float16)feat[rl * 3 * ND + ND + sub];
                    a.FqH[omds_screen_fidx(2 * d + sub, t, a.ldF)] = (_Float16)feat[rl * 3 * ND + 2 * ND + sub];
                    if (m.scrQ) {   // skip-connection networks: the concatenation operand of the screening kernel
                        _Float16* S = reinterpret_cast<_Float16*>(m.scrQ);
                        S[omds_screen_sidx(sub, 3 * d, t, a.ldF)] = (_Float16)v;
                        S[omds_screen_sidx(d + sub, 3 * d, t, a.ldF)] = (_Float16)feat[rl * 3 * ND + ND + sub];
                        S[omds_screen_sidx(2 * d + sub, 3 * d, t, a.ldF)] = (_Float16)feat[rl * 3 * ND + 2 * ND + sub];
                    }
                }
            }
        }
    }
    OMDS_TL_STAMP(10);
    if (a.st.step >= a.st.H) return;   // last step: nothing is integrated, no next network evaluation
    __syncthreads();

    // ---- the encoded joint inputs of the next step (as k_rollout_features writes them) --------
    {
        const int d = m.d;
        for (int e = tid; e < RW * 3 * ND; e += P2_NT) {
            const int rl = e / (3 * ND), cc = e - rl * (3 * ND), part = cc / ND, t = t_base + rl;
            if (t < t_end) a.FqOut[(size_t)t * OMDS_FROW + part * d + (cc - part * ND)] = feat[e];
        }
    }
    OMDS_TL_STAMP(11);
    OMDS_TL_STAMP(19);
}

#ifdef OMDS_TAIL_TL
// one line per workgroup: XCC / CU of thread 0 are not recorded; columns = stamps 0..11 and 19 (0 and 19: s_memrealtime, 100 MHz)
__global__ void k_tail_tl_dump(int nb) {
    for (int b = 0; b < nb && b < 1024; ++b) {
        printf("TL %d", b);
        for (int i = 0; i < 19; ++i) printf(" %llu", g_tail_tl[b][i]);   // 12..16: inside modulate_core, 17..18: pass2_body_g4
        printf(" %llu\n", g_tail_tl[b][19]);
    }
}
#endif

template <int ND, int ROWS, int ACT = OMDS_ACT_RELU>
static void launch_tail_sel_t(hipStream_t s, const TailArgs& a) {
    static std::atomic<uint64_t> configured{0};
    const size_t extra = 32 * 4;   // sel
    if (omds_first_use_on_device(configured)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tail_sel<ND, ROWS, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(tail_lds_bytes(OMDS_MAX_HIDDEN + 1) + extra));
    }
    const int RW = (ROWS == 4 ? 20 : ROWS) / a.st.k;
    hipLaunchKernelGGL((k_tail_sel<ND, ROWS, ACT>), dim3((a.t_end - a.t_begin + RW - 1) / RW), dim3(P2_NT), tail_lds_bytes(a.m.nhh + 1) + extra, s, a);
#ifdef OMDS_TAIL_TL
    {
        static int seen = 0;
        static const int want = OMDS_EXP_ENV("OMDS_TAIL_TL_STEP", -1);
        if (a.st.step == want && ++seen == 3) hipLaunchKernelGGL(k_tail_tl_dump, dim3(1), dim3(1), 0, s, (a.t_end - a.t_begin + RW - 1) / RW);
    }
#endif
}

// 16-row tiles (bit-identical to 32-row ones, mlp_device.h) while their workgroups still fit the CUs two at a time: twice as
// many, half as long, and the second resident fills the first one's top-k / gather / modulation phases
// Tile shapes forced by the test hook omds_debug_force_tile_rows (libomds_hip_test.so, include/omds_test.h) or, in experiment
// builds, by the environment (OMDS_TAIL_SEL_ROWS = 4 | 16 | 32, OMDS_TAIL_ROWS = 16 | 32): every shape computes the same bits, and
// the tests say so by running them against each other.  The release library has neither: 0 = the launcher's own choice.
#ifdef OMDS_TEST_HOOKS
static std::atomic<int> g_force_sel_rows{0}, g_force_tail_rows{0};
void omds_force_tile_rows(int tail_sel_rows, int tail_rows) {
    g_force_sel_rows.store(tail_sel_rows);
    g_force_tail_rows.store(tail_rows);
}
#define OMDS_FORCED_ROWS(slot, env) ((slot).load() > 0 ? (slot).load() : OMDS_EXP_ENV(env, 0))
#else
#define OMDS_FORCED_ROWS(slot, env) OMDS_EXP_ENV(env, 0)
#endif
static int tail_sel_rows(int N, int k, bool g4_ok) {
    const int forced = OMDS_FORCED_ROWS(g_force_sel_rows, "OMDS_TAIL_SEL_ROWS");
    if (k > 16) return 32;
    if (forced == 4 && g4_ok && k <= 20) return 4;
    if (forced == 16 || forced == 32) return forced;
    // 4-row groups (20 rows per workgroup) where the busiest CU multiplies less that way: matrix-pipe cycles per layer of the CU
    // with the most workgroups, 5 groups x 2048 (x 1.1: one wave per SIMD issues a 4x4x1 every 9 cycles, not 8) against 8192 per
    // 16-row tile.  N = 1024, k = 5: 256 workgroups, one per CU, against 342 tiles of 16 rows (40.5 -> 35 us); N = 4096: four
    // per CU against six tiles (20.5 -> 20.1 ms per iteration); N = 1500: 375 workgroups would put two on 119 CUs where 500
    // tiles of 16 rows are two per CU and cheaper.  A lone 16-row tile (no CU with two) is the shorter chain anyway.
    const int RW16 = 16 / k;
    if (forced == 0 && g4_ok && k <= 10) {
        const int ncu = omds_cu_count(), RW4 = 20 / k;
        const long long wg16 = (N + RW16 - 1) / RW16, wg4 = (N + RW4 - 1) / RW4;
        const long long cost16 = ((wg16 + ncu - 1) / ncu) * 8192, cost4 = ((wg4 + ncu - 1) / ncu) * (5 * 2048 * 11 / 10);
        if (wg16 > ncu && cost4 < cost16) return 4;
    }
    return (N + RW16 - 1) / RW16 <= 512 ? 16 : 32;
}

bool omds_tail_sel_supported(int n_dof, int k) { return (n_dof == 7 || n_dof == 2) && k >= 1 && k <= 32; }

void omds_launch_tail_sel(hipStream_t s, const MlpDev& m, const float* Fp, const float* radius, const float* xyzr,
                          float* Fq, int O, const StepArgs& st, const int* rowlist, const int* range, const ExactOut& ex,
                          uint16_t* FqH, int ldF, float e_bound, unsigned* viol) {
    TailArgs a;
    a.e_bound = e_bound;
    a.viol = viol;
    a.FqH = reinterpret_cast<_Float16*>(FqH);
    a.ldF = ldF;
    a.t_begin = 0; a.t_end = st.N;
    a.slot0 = 0; a.n_slots = 0;
    static const int stop = OMDS_EXP_ENV("OMDS_TAIL_SEL_STOP", 0);
    a.dbg_stop = stop;
    a.m = m; a.Fp = Fp; a.radius = radius; a.xyzr = xyzr; a.Dmin = nullptr; a.Fq = Fq; a.FqOut = Fq; a.dscr = nullptr; a.O = O; a.st = st;
    a.rowlist = rowlist; a.range = range; a.ex = ex;
    if (m.act == OMDS_ACT_TANH) {   // derivative rows instead of masks: 16- or 32-row tiles (the 4-row-group backward is a ReLU-mask form)
        const int rows = tail_sel_rows(st.N, st.k, false);
        if (st.n == 7) { if (rows == 16) launch_tail_sel_t<7, 16, OMDS_ACT_TANH>(s, a); else launch_tail_sel_t<7, 32, OMDS_ACT_TANH>(s, a); }
        else { if (rows == 16) launch_tail_sel_t<2, 16, OMDS_ACT_TANH>(s, a); else launch_tail_sel_t<2, 32, OMDS_ACT_TANH>(s, a); }
        return;
    }
    const int rows = tail_sel_rows(st.N, st.k, m.skip_mask == 0 && m.nhh >= 1);
    if (st.n == 7) { if (rows == 4) launch_tail_sel_t<7, 4>(s, a); else if (rows == 16) launch_tail_sel_t<7, 16>(s, a); else launch_tail_sel_t<7, 32>(s, a); }
    else { if (rows == 4) launch_tail_sel_t<2, 4>(s, a); else if (rows == 16) launch_tail_sel_t<2, 16>(s, a); else launch_tail_sel_t<2, 32>(s, a); }
}

static size_t tail_lds_bytes(int nhid) {
    return ((size_t)P2_MT * LDH + 32 * 33) * 4 + (size_t)nhid * P2_NT * 4 + 3 * P2_MT * 4 +
           (32 * 12 + 32 + 32 * 3 * OMDS_MAX_DOF) * 4;
}

template <int ND, int ACT, int ROWS>
static void launch_tail_a(hipStream_t s, const TailArgs& a) {
    static std::atomic<uint64_t> configured{0};
    if (omds_first_use_on_device(configured)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tail<ND, ACT, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)tail_lds_bytes(OMDS_MAX_HIDDEN + 1));
    }
    const int RW = (ROWS == 4 ? 20 : ROWS) / a.st.k;
    hipLaunchKernelGGL((k_tail<ND, ACT, ROWS>), dim3((a.t_end - a.t_begin + RW - 1) / RW), dim3(P2_NT), tail_lds_bytes(a.m.nhh + 1), s, a);
#ifdef OMDS_TAIL_TL
    {
        static int seen = 0;
        static const int want = OMDS_EXP_ENV("OMDS_TAIL_TL_STEP", -1);
        if (a.st.step == want && ++seen == 3) hipLaunchKernelGGL(k_tail_tl_dump, dim3(1), dim3(1), 0, s, (a.t_end - a.t_begin + RW - 1) / RW);
    }
#endif
}

template <int ND, int ROWS>
static void launch_tail_t(hipStream_t s, const TailArgs& a) {
    if (a.m.act == OMDS_ACT_RELU) launch_tail_a<ND, OMDS_ACT_RELU, ROWS>(s, a);
    else if constexpr (ROWS != 4) launch_tail_a<ND, OMDS_ACT_TANH, ROWS>(s, a);
}

bool omds_tail_supported(int n_dof, int k) { return (n_dof == 7 || n_dof == 2) && k >= 1 && k <= P2_MT; }

// 16-row tiles when 32-row tiles would leave most CUs without a workgroup (OMDS_TAIL_ROWS=4|16|32 forces one); g4_ok (ReLU network
// without skip concatenations): 4-row groups, 20 rows per workgroup (k_tail<ND, ACT, 4>), while its workgroups fit the CUs in ONE
// round -- 5 groups x 2048 x 1.1 matrix-pipe cycles per hidden layer against 16 384 per 32-row tile.  N = 1024, k = 5: 256 workgroups,
// one per CU, against 171 tiles of 32 rows (71.6 -> 58.4 us).  Beyond one round the 32-row tile wins: the 4-row-group kernel holds
// the weight ring in 64 registers (134 in all: one workgroup per CU, nothing fills its top-k / modulation phases) and multiplies on
// four of its eight waves -- N = 4096: 233 against 200 us, N = 8192: 460 against 358 (tools/tail_rows_ab.sh).
int omds_tail_rows(int N, int k, bool g4_ok) {
    const int forced = OMDS_FORCED_ROWS(g_force_tail_rows, "OMDS_TAIL_ROWS");
    if (k > 16) return 32;
    if (forced == 4 && g4_ok && k <= 20) return 4;
    if (forced == 16 || forced == 32) return forced;
    const int RW32 = 32 / k;
    const long long wg32 = (N + RW32 - 1) / RW32;
    if (wg32 <= 128) return 16;
    if (forced == 0 && g4_ok && k <= 10) {
        const int ncu = omds_cu_count(), RW4 = 20 / k;
        const long long wg4 = (N + RW4 - 1) / RW4;
        if (wg4 <= ncu) return 4;
    }
    return 32;
}

int omds_tail_scratch_rows(int N, int k) {
    const int RW32 = P2_MT / k;
    int rows = (N + RW32 - 1) / RW32 * 32;
    if (k <= 16) { const int RW16 = 16 / k; rows = std::max(rows, (N + RW16 - 1) / RW16 * 16); }
    return rows;
}

void omds_launch_tail(hipStream_t s, const MlpDev& m, const float* Fp, const float* radius, const float* xyzr,
                      const float* Dmin, float* Fq, float* dscr, int O, const StepArgs& st, int t_begin, int t_end, uint16_t* FqH, int ldF,
                      float* FqOut, const int* guard_range, float e_bound, unsigned* viol) {
    TailArgs a;
    a.FqH = reinterpret_cast<_Float16*>(FqH);
    a.ldF = ldF;
    const int rows = omds_tail_rows(st.N, st.k, m.act == OMDS_ACT_RELU && m.skip_mask == 0 && m.nhh >= 1);
    const int RW = (rows == 4 ? 20 : rows) / st.k;
    a.t_begin = t_begin; a.t_end = t_end;          // t_begin must be a multiple of RW
    a.slot0 = t_begin / RW;
    a.n_slots = (st.N + RW - 1) / RW;
    static const int stop = OMDS_EXP_ENV("OMDS_TAIL_STOP", 0);
    a.dbg_stop = stop;
    a.rowlist = nullptr; a.range = guard_range; a.ex = ExactOut{}; a.e_bound = e_bound; a.viol = viol;
    a.m = m; a.Fp = Fp; a.radius = radius; a.xyzr = xyzr; a.Dmin = Dmin; a.Fq = Fq; a.FqOut = FqOut ? FqOut : Fq; a.dscr = dscr; a.O = O; a.st = st;
    if (rows == 4) {
        if (st.n == 7) launch_tail_t<7, 4>(s, a);
        else launch_tail_t<2, 4>(s, a);
    } else if (rows == 16) {
        if (st.n == 7) launch_tail_t<7, 16>(s, a);
        else launch_tail_t<2, 16>(s, a);
    } else {
        if (st.n == 7) launch_tail_t<7, 32>(s, a);
        else launch_tail_t<2, 32>(s, a);
    }
}

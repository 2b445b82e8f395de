// Low-precision SCREENING of pass 1 with exact re-selection (gfx950).
//
// Pass 1 of MPPI.distance_repulsion_nn (MPPI.py:233-253) evaluates the distance network on all N*O (rollout, obstacle)
// pairs, but only feeds the sort that picks the k closest obstacles; the distance and gradient the step uses come from
// pass 2 on those k rows (MPPI.py:259-280).  So the N*O evaluations need not be fp32 as long as the SELECTED SET is the
// fp32 one.  Three kernels replace k_pass1 in the horizon step:
//
//   k_screen  : the network on all pairs in fp16 (v_mfma_f32_32x32x16_f16, fp32 accumulate) -> approximate min link
//               distance Da[t][o].
//   k_select  : per rollout, tau = (k-th smallest Da) + 1.25 eps; every obstacle with Da <= tau is a CANDIDATE.  If the rows
//               that are NOT candidates have a screening error Da - D <= eps and the exact k-th smallest candidate stays eps
//               below tau (the slack guard k_tail_sel checks from exact numbers), the exact top-k over the candidates is the
//               exact top-k over all obstacles, ties included (proof in DESIGN.md 4.3).  The candidates of a rollout take a
//               contiguous range of a compact list (range start, length and tau per rollout).
//   k_exact   : the fp32 pass-1 tile code (pass1_tile, bit-identical arithmetic per row: an MFMA output element is one
//               k-ordered fmaf chain of its own row) on the listed rows only.  Per entry it leaves the exact D and what
//               pass 2's forward would compute for the row (same arithmetic): pass-2 distance, arg-min link, ReLU masks --
//               k_tail_sel then picks the top-k by (D, obstacle) and runs only the backward.  It also records
//               max |Da - D| over all candidates (the run-time guard of eps: the host re-runs the propagate without
//               screening if the observed error ever exceeds the calibrated margin).
//
// k_screen design (MI355X-first).  fp16 MFMA is 16x the fp32 rate, so everything the fp32 kernel could hide -- LDS round
// trips of the activations, weight fragments from L2 -- would dominate.  Hence:
//   * TRANSPOSED product H'^T = W . H^T: the weights are the A operand, the activations the B operand (column = pair).
//     A wave owns 32 pairs and ALL 256 features of them, so a layer's output (C layout: lane = pair, registers =
//     features) IS the next layer's B operand after ReLU + cvt_pk -- the K order inside a dot product is free, the host
//     packs the weights in the permuted order the C layout produces.  Activations never leave registers: no LDS traffic,
//     no bank conflicts, no barrier for them.
//   * weights stream L2 -> LDS by LDS-DMA (buffer/global_load ... lds, 1 KiB = one A fragment per wave instruction) as
//     16 KB slices (32 output features x 256 k) through a 4-slot ring shared by the 8 waves of the workgroup (256 pairs):
//     128 KB per layer per 256 pairs instead of per 32, one s_barrier per slice, counted vmcnt, loads two slices ahead.
//   * layer 1 runs on the matrix pipe too: its 3(n+3) <= 32 inputs are two k-chunks, fetched as fp16 from a 64-byte row per
//     rollout and per obstacle (four 16-byte loads per lane and tile; reading the separable fp32 halves Fq[t] + Fp[o]
//     like k_pass1 does cost 36 % of the kernel, their fp16 copies still 25 %), W1 is one more slice of the ring.
//   * persistent workgroups (one per CU) loop over their tiles: the ring never drains, no refill gap between tiles.
//   * ReLU networks: a k-chunk (16 hidden units) whose activations are zero for all 32 pairs of the wave adds nothing to any
//     output, so its MFMAs are not issued (a ballot per tested chunk and layer, a scalar branch per MFMA of chunks 8-15).  The
//     host orders the units of this pack by how often they fire (capi.hip: screen_reorder), which puts the units a trained
//     network never uses -- 300 of the shipped one's 1024 -- into whole chunks: a quarter of the MFMAs go (EXPERIMENTS.md C 4.1d).
#include <algorithm>
#include <cstdio>
#include <vector>

#include "mlp_device.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// Geometry: SC_WAVES waves per workgroup, each owning SC_RB blocks of 32 pairs; every A fragment (1 KiB of weights) a wave
// reads from the ring feeds SC_RB MFMAs.  Default 8 waves x 1 block (two waves per SIMD, 241 registers).  The alternative
// 4 waves x 2 blocks (one wave per SIMD with the 512-register budget: half the LDS reads per MFMA; make variant V=rb2
// VFLAGS="-DOMDS_SC_WAVES=4 -DOMDS_SC_RB=2") is bit-identical and measured 104 vs 99 us per launch: the LDS port is not the
// limit (tools/ubench/mfma_f16_issue.hip: one fragment read per MFMA sustains 83-87 % of the fp16 MFMA peak at two waves
// per SIMD, 63-65 % at one), a single wave per SIMD hides less.
#ifndef OMDS_SC_WAVES
#define OMDS_SC_WAVES 8
#endif
#ifndef OMDS_SC_RB
#define OMDS_SC_RB 1
#endif
#ifndef OMDS_SC_RING
#define OMDS_SC_RING 4
#define OMDS_SC_DIST 3
#endif
constexpr int SC_WAVES = OMDS_SC_WAVES;
constexpr int SC_RB = OMDS_SC_RB;             // 32-pair blocks per wave
constexpr int SC_NT = SC_WAVES * 64;
constexpr int SC_ROWS = SC_WAVES * SC_RB * 32; // pairs per workgroup
constexpr int SC_SLICE = 16384;               // bytes: 16 k-chunks x 1 KiB fragment
constexpr int SC_RING = OMDS_SC_RING;         // ring slots of 16 KB (a power of two)
constexpr int SC_DIST = OMDS_SC_DIST;         // slices in flight ahead of the one being multiplied (<= RING - 1)
constexpr int SC_PW = 16 / SC_WAVES;          // LDS-DMA pieces (1 KiB fragments) per wave and slice
constexpr int SC_MAX_TILES = 20;              // tiles per workgroup whose results fit the LDS next to the ring
#ifndef OMDS_SC_TEST0
#define OMDS_SC_TEST0 10
#endif
// first k-chunk whose activations are tested for "all zero" before its MFMAs (ReLU networks).  Measured on the shipped network
// (tools/studies/screen_test0_sweep.sh, 1024 x 32 / 4096 x 32): 4: 100.0 / 365.5 us, 8: 98.0 / 357.7, 9: 96.7 / 351.2, 10: 96.1 / 349.5,
// 11: 97.0 / 351.7, 12: 97.2 / 355.3 -- its silent units begin at chunk 9-13 depending on the layer; every test and branch in front of
// them costs issue slots
constexpr int SC_TEST0 = OMDS_SC_TEST0;

struct ScreenArgs {
    const _Float16* Wh;      // [nhh*8 + 2 slices][16 fragments][64 lane][8 halfs], fragment order (packed by omds_set_mlp)
    const float* bias;       // [nhh + 2][256]: layer 1, hidden->hidden, the last layer's (padded with zeros)
    const _Float16* FqH;     // [4 pieces][ldFq rows][8]: fp16 network inputs of the rollouts (q, sin q, cos q at their feature slots)
    const _Float16* FpH;     // [4][ldFp][8]: obstacle points likewise (omds_screen_fidx)
    const _Float16* FqS;     // skip-connection networks: the same inputs at the slots of the CONCATENATED columns (omds_screen_sidx)
    const _Float16* FpS;
    int ldFq, ldFp;          // row capacities of the tables
    const float* radius;
    float* Dmin;
    long long total_rows;
    // workgroup w owns units_base + (w < units_rem) consecutive UNITS of `unit` pairs (a unit = a 256-pair tile, or -- when the
    // flush phase selects -- a whole rollout of O pairs), starting at unit w * units_base + min(w, units_rem)
    int unit, units_base, units_rem;
    int B, O;
    uint32_t ignored;
    uint32_t skip_mask;      // bit L: the encoded input is concatenated behind level L (MlpDev::skip_mask)
    OmdsDivisor odiv;
    int nhh, C;
    float out_div;
    int res_tiles;           // tiles per workgroup (capacity of the result buffer in LDS)
    const SelectSink* sel;   // device memory, or nullptr.  Non-null: the flush phase selects each rollout's candidates instead of writing
                             // Dmin.  A pointer on purpose: as kernel arguments the sink's 20 dwords were loaded at entry and held in
                             // SGPRs across the tile loop (SGPR spills 36 -> 114)
    unsigned long long* tl;  // diagnostic (OMDS_SCREEN_TL=1): [workgroup][8] s_memtime stamps, nullptr otherwise
    int dbg;                 // timing experiments only (OMDS_SCREEN_DBG): 1 = no weight streaming after the prologue,
                             // 2 = no layer-1 loads, 4 = no per-slice wait + barrier
};

// lowbias32 (an integer hash with good avalanche): which non-candidate pairs enter the audit sample
__device__ __forceinline__ unsigned omds_audit_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// LDS-DMA: 16 bytes per lane from gsrc (per-lane address) to LDS [lds_dst + 16 * lane] (lds_dst wave-uniform).  Invisible
// to hipcc's s_waitcnt bookkeeping by design: completion is waited for with counted vmcnt below.
// The source is a wave-uniform base (SGPR pair) + a per-lane 32-bit byte offset + an immediate: per-lane 64-bit
// pointers would be loop-invariant VGPR pairs, one per slice, that hipcc hoists out of the tile loop and spills.
template <int IMM>
__device__ __forceinline__ void dma16(const void* gbase_uniform, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(gbase_uniform), "s"(lds_dst), "n"(IMM)
                 : "memory");
}
// counted wait for this wave's LDS-DMA pieces + workgroup barrier; the "memory" clobber keeps hipcc from moving LDS reads
// of the slice above the barrier
template <int N>
__device__ __forceinline__ void wait_vm_barrier_n() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory"); }
// n = LDS-DMA pieces of this wave that may stay in flight (a compile-time constant once the slice loop is unrolled)
__device__ __forceinline__ void wait_vm_barrier(int n) {
    switch (n) {
        case 0: wait_vm_barrier_n<0>(); break;
        case 2: wait_vm_barrier_n<2>(); break;
        case 4: wait_vm_barrier_n<4>(); break;
        case 6: wait_vm_barrier_n<6>(); break;
        case 8: wait_vm_barrier_n<8>(); break;
        case 10: wait_vm_barrier_n<10>(); break;
        case 12: wait_vm_barrier_n<12>(); break;
        case 16: wait_vm_barrier_n<16>(); break;
        case 20: wait_vm_barrier_n<20>(); break;
        default: wait_vm_barrier_n<0>(); break;
    }
}

// two fp32 pre-activations -> two fp16 activations (the next layer's B operand).  ReLU: convert, then one packed max.
// tanh: 1 - 2 / (1 + exp(2x)) in fp32 (v_exp_f32 / v_rcp_f32: +inf and 0 give the saturated values, tanh(0) = 0 exactly), then
// convert -- 7 VALU instructions per pair against 2 (EXPERIMENTS.md C 4.1b has what that costs beside the MFMAs)
template <int ACT>
__device__ __forceinline__ h2 act_pk(float a, float b) {
    if constexpr (ACT == OMDS_ACT_RELU) {
        h2 p = {(_Float16)a, (_Float16)b};                 // v_cvt_pk_f16_f32 (round to nearest even)
        const h2 z = {(_Float16)0, (_Float16)0};
        return __builtin_elementwise_max(p, z);            // v_pk_max_f16
    } else {
        // a, b arrive as 2 log2(e) x: the host scales the fp16 weights and the biases of every tanh layer by that factor
        // (build_mlp_packs, OMDS_SCREEN_TANH_SCALE), so tanh x = 1 - 2 / (1 + exp2(a)) needs no multiply here.  The add and the final
        // multiply-add are packed fp32 instructions (v_pk_add_f32 / v_pk_fma_f32: one per pair), only exp and rcp run per element:
        // 7 VALU instructions per pair (round 3: 11 -- scalar mul / add / fma; k_screen<2, tanh> 402 -> 382 us with the packed forms)
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 e1 = f2{__builtin_amdgcn_exp2f(a), __builtin_amdgcn_exp2f(b)} + 1.f;
        const f2 r = f2{__builtin_amdgcn_rcpf(e1[0]), __builtin_amdgcn_rcpf(e1[1])};
        const f2 t = __builtin_elementwise_fma(f2{-2.f, -2.f}, r, f2{1.f, 1.f});
        return h2{(_Float16)t[0], (_Float16)t[1]};
    }
}

// A fragments of one group (4 k-chunks) of a slice
struct AGroup { h8 f[4]; };
__device__ __forceinline__ AGroup read_group(const unsigned char* slot_lane, int g) {
    AGroup r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.f[i] = *reinterpret_cast<const h8*>(slot_lane + (4 * g + i) * 1024);
    return r;
}

#define SC_TL(i) do { if (a.tl && threadIdx.x == 0 && it == 0) a.tl[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)

// pair (rollout t, obstacle o) of this lane in tile `tile` of the workgroup's chunk [p0, p1); rows past the end (also: the
// prefetch of a non-existent next tile) clamp to the chunk's last pair and are not stored.  Plain scalars on purpose: a
// struct carried around the tile loop went through scratch memory
__device__ __forceinline__ void tile_row(const ScreenArgs& a, long long p0, long long p1, int tile, int wave, int rb, int b, unsigned& t, unsigned& o) {
    long long row = p0 + (long long)tile * SC_ROWS + (wave * SC_RB + rb) * 32 + b;
    if (row >= p1) row = p1 - 1;
    t = a.odiv.div((unsigned)row);
    o = (unsigned)row - t * (unsigned)a.O;
}

typedef unsigned u4 __attribute__((ext_vector_type(4)));

// wave-wide minimum of a float, result uniform over the wave (NaN never wins: v_min_f32 returns the other operand): four DPP
// steps inside each row of 16 lanes (the patterns of pass1_tile's link minimum), then the four rows meet through SGPRs --
// ~10 instructions where a __shfl_xor butterfly is 6 dependent ds_bpermute round trips
__device__ __forceinline__ float wave_min_f32(float v) {
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)));
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)));
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}
// number of set bits of a wave mask below this lane
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// One wave = one rollout: from the rollout's O screening values (row: global memory or LDS) to its candidates in the list
// and its share of the audit sample.  All SEL_WAVES waves of the workgroup call it together (two barriers inside: ONE atomic on
// the list counter per workgroup -- 1024 same-address atomics, one per rollout, serialise in L2 and were half of the first
// k_select); `live` = this wave has a rollout.  Everything cross-lane is a ballot or a DPP step: this runs in the flush phase
// of k_screen, on the critical path of every horizon step, and a shuffle-based version took 9 us there.
constexpr int SEL_WAVES = 8;
struct SelectShared { int wtot[SEL_WAVES], watot[SEL_WAVES], wbase, wabase; };
template <typename RowPtr>
__device__ __forceinline__ void select_rollout(const SelectSink& a, SelectShared& sh, RowPtr row, int t, bool live, int O, int lane, int wave) {
    constexpr int NV = 8;
    float v[NV];
    const bool in_regs = O <= 64 * NV;   // rows of up to 512 values sit in registers; longer rows are re-read
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int o = lane + 64 * i; v[i] = (o < O) ? row[o] : __builtin_nanf(""); }   // NaN padding never wins a minimum
    }
    // k-th smallest value, multiplicities counted: at most k rounds of "smallest value above the previous one, and how many
    // entries hold it".  NaN entries (an fp16 overflow inside the network) never win a round; they are made candidates below.
    float kth = __builtin_inff(), prev = 0.f;
    int remaining = a.k;
    for (int round = 0; round < a.k; ++round) {
        float loc = __builtin_inff();
        if (in_regs) {
#pragma unroll
            for (int i = 0; i < NV; ++i) loc = (round == 0 || v[i] > prev) ? fminf(loc, v[i]) : loc;
        } else {
            for (int o = lane; o < O; o += 64) { const float x = row[o]; loc = (round == 0 || x > prev) ? fminf(loc, x) : loc; }
        }
        const float m = wave_min_f32(loc);
        int c = 0;
        if (in_regs) {
#pragma unroll
            for (int i = 0; i < NV; ++i) c += __popcll(__ballot(v[i] == m));
        } else {
            for (int o0 = 0; o0 < O; o0 += 64) { const int o = o0 + lane; c += __popcll(__ballot(o < O && row[o] == m)); }
        }
        kth = m;
        if (c >= remaining || !(m < __builtin_inff())) break;   // +inf: fewer than k finite values -- everything becomes a candidate
        remaining -= c;
        prev = m;
    }
    const float tau = kth + a.delta;
    // Candidates: everything not above tau, and every non-finite value (NaN / +inf = an fp16 overflow inside the network:
    // nothing is known about such a row).  AUDIT sample: a pseudo-random subset of the pairs that are NOT candidates -- the
    // population the selection rule makes an assumption about (screening error <= eps) -- goes with its screening value into
    // the propagate's audit list; k_audit evaluates those pairs in fp32 once the horizon loop is done.
    const int rbase = t * O;
    const bool auditing = a.audit_mask != 0xffffffffu;
    auto is_cand = [&](float x) { return !(x > tau) || !(x < __builtin_inff()); };
    auto is_audit = [&](int o) { return auditing && (omds_audit_hash((unsigned)(rbase + o) ^ a.audit_seed) & a.audit_mask) == 0u; };
    // one ballot pair per group of 64 obstacles: totals are scalar popcounts, a lane's list position is the number of set bits
    // below it -- no scans, no shuffles.  (The order of a rollout's entries in the list is group-major; nothing depends on it.)
    unsigned long long cm[NV], am[NV];
    int wave_total = 0, wave_audit = 0;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int o = lane + 64 * i;
            const bool c = o < O && is_cand(v[i]);
            cm[i] = __ballot(c);
            am[i] = __ballot(o < O && !c && is_audit(o));
            wave_total += __popcll(cm[i]);
            wave_audit += __popcll(am[i]);
        }
    } else {
        for (int o0 = 0; o0 < O; o0 += 64) {
            const int o = o0 + lane;
            const bool c = o < O && is_cand(row[o]);
            wave_total += __popcll(__ballot(c));
            wave_audit += __popcll(__ballot(o < O && !c && is_audit(o)));
        }
    }
    if (!live) { wave_total = 0; wave_audit = 0; }
    if (lane == 0) { sh.wtot[wave] = wave_total; sh.watot[wave] = wave_audit; }
    __syncthreads();
    // the two list counters are bumped by two different waves at once: their L2 round trips overlap
    if (threadIdx.x == 0) {
        int sum = 0;
#pragma unroll
        for (int w = 0; w < SEL_WAVES; ++w) sum += sh.wtot[w];
        sh.wbase = atomicAdd(a.total, sum);
    } else if (threadIdx.x == 64) {
        int asum = 0;
#pragma unroll
        for (int w = 0; w < SEL_WAVES; ++w) asum += sh.watot[w];
        sh.wabase = asum ? atomicAdd(a.audit_total, asum) : 0;
    }
    __syncthreads();
    if (live) {
        int base = sh.wbase, abase = sh.wabase;
        for (int w = 0; w < wave; ++w) { base += sh.wtot[w]; abase += sh.watot[w]; }
        if (lane == 0) {
            a.range[4 * t] = base;
            a.range[4 * t + 1] = wave_total;
            a.range[4 * t + 2] = __builtin_bit_cast(int, tau);   // k_tail_sel checks the rollout's slack against it
        }
        const int arow0 = (a.step_row0 + t) * O;
        if (in_regs) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int o = lane + 64 * i;
                if ((cm[i] >> lane) & 1ull) {
                    const int pos = base + lanes_below(cm[i]);
                    a.rowlist[pos] = rbase + o;
                    if (a.listDa) a.listDa[pos] = v[i];
                } else if ((am[i] >> lane) & 1ull) {
                    const int apos = abase + lanes_below(am[i]);
                    if (apos < a.audit_cap) { a.audit_rows[apos] = arow0 + o; a.audit_da[apos] = v[i]; }
                }
                base += __popcll(cm[i]);
                abase += __popcll(am[i]);
            }
        } else {
            for (int o0 = 0; o0 < O; o0 += 64) {
                const int o = o0 + lane;
                const float x = o < O ? row[o] : 0.f;
                const bool c = o < O && is_cand(x), au = o < O && !c && is_audit(o);
                const unsigned long long cmk = __ballot(c), amk = __ballot(au);
                if (c) {
                    const int pos = base + lanes_below(cmk);
                    a.rowlist[pos] = rbase + o;
                    if (a.listDa) a.listDa[pos] = x;
                } else if (au) {
                    const int apos = abase + lanes_below(amk);
                    if (apos < a.audit_cap) { a.audit_rows[apos] = arow0 + o; a.audit_da[apos] = x; }
                }
                base += __popcll(cmk);
                abase += __popcll(amk);
            }
        }
    }
    __syncthreads();   // the shared counters are reused by the workgroup's next round of rollouts
}

template <int NHH, int ACT, bool SKIP>
__global__ __launch_bounds__(SC_NT, SC_WAVES == 4 ? 1 : 2) void k_screen(ScreenArgs a) {
    // PERSISTENT: workgroup w multiplies the tiles of its own contiguous chunk of the pair space (one workgroup per CU).
    // The weight slices keep streaming through the ring across tile boundaries (the slice sequence is periodic), the bias
    // table is loaded once, and the results wait in LDS until the end -- a store in flight would perturb the counted vmcnt
    // waits of the LDS-DMA pieces (stores and loads retire out of order with respect to each other).  Measured on the
    // one-tile-per-workgroup form (OMDS_SCREEN_TL): of 57 kcycles per tile slot 14 were the refill gap between two
    // workgroups of a CU, 12 the layer-1 operand loads and first-slice latency, and only 29 the slice loop.
    // A chunk of WHOLE rollouts (a.sel.rowlist != nullptr) ends with every screening value of those rollouts in this
    // workgroup's LDS: the flush phase then does k_select's work from there -- k-th smallest, window, list append, audit
    // sample -- and the N x O matrix is never written or re-read (one launch and 2 x 1.2 MB less per horizon step).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // the small tables first: their ds_read offsets then fit the 16-bit immediate from ONE base register (behind the 64 KB
    // ring hipcc kept a base VGPR per 256-byte window alive across the tile loop, and spilled them)
    float* biasL = reinterpret_cast<float*>(smem_raw);                         // [NHH+2][256]: layer 1, hidden->hidden, last
    float* resL = biasL + (NHH + 2) * OMDS_WIDTH;                              // [a.res_tiles][SC_ROWS]
    unsigned char* ring = reinterpret_cast<unsigned char*>(resL + a.res_tiles * SC_ROWS);   // [SC_RING][SC_SLICE]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane & 31, half = lane >> 5;
    constexpr int S = NHH * 8 + 2;                 // slice steps of a tile: layer 1, 8 per hidden->hidden layer, the last layer
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const int wg = blockIdx.x;                                                  // this workgroup's pairs [p0, p1)
    const long long p0 = ((long long)wg * a.units_base + (wg < a.units_rem ? wg : a.units_rem)) * a.unit;
    const long long p1e = p0 + (long long)(a.units_base + (wg < a.units_rem ? 1 : 0)) * a.unit;
    const long long p1 = p1e < a.total_rows ? p1e : a.total_rows;
    const int my_pairs = (int)(p1 - p0);
#ifdef OMDS_SC_EXPERIMENT   // OMDS_SCREEN_DBG 16: the partial last tile of a chunk does not exist at all (neither its MFMAs nor its ring traffic)
    const int my_tiles = (my_pairs + SC_ROWS - 1) / SC_ROWS - (((OMDS_DBG(a.dbg) & 16) && (my_pairs % SC_ROWS) != 0 && my_pairs > SC_ROWS) ? 1 : 0);
#else
    const int my_tiles = (my_pairs + SC_ROWS - 1) / SC_ROWS;                    // >= 1: the grid never exceeds the chunks
#endif

    // slice `sl` of the network -> ring slot `slot`; this wave moves fragments PW*w .. PW*w + PW-1
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(a.Wh) + (SC_PW * wave) * 1024;   // wave-uniform
    const unsigned lane16 = lane * 16;
    auto issue = [&](int sl, int slot) {
        const unsigned char* src = wbase + (size_t)sl * SC_SLICE;
        const unsigned dst = ring_lds + (unsigned)(slot * SC_SLICE + (SC_PW * wave) * 1024);
        // the instruction offset advances the global AND the LDS address
        dma16<0>(src, lane16, dst);
        if (SC_PW > 1) dma16<1024>(src, lane16, dst);
        if (SC_PW > 2) { dma16<2048>(src, lane16, dst); dma16<3072>(src, lane16, dst); }
    };
    if (a.tl && threadIdx.x == 0) { a.tl[(size_t)blockIdx.x * 8 + 7] = __builtin_readcyclecounter(); a.tl[(size_t)blockIdx.x * 8 + 5] = wall_clock64(); }
    for (int i = tid; i < (NHH + 2) * OMDS_WIDTH; i += SC_NT) biasL[i] = a.bias[i];
    for (int i = tid; i < my_tiles * SC_ROWS; i += SC_NT) resL[i] = __builtin_inff();   // the two lane-halves of a pair min into it
    __syncthreads();   // bias table visible; nothing of the ring is in flight yet (hipcc's fence would drain it)
#pragma unroll
    for (int s0 = 0; s0 < SC_DIST; ++s0) issue(s0 % S, s0 % SC_RING);

    // network inputs of a pair as fp16 B operands: chunk cc, lane-half h, slot j <-> feature 16 cc + 8 h + j of
    // [x, sin x, cos x] (x = q then the obstacle point).  The rollout table holds the q features (zeros elsewhere), the
    // obstacle table the point features: two 16-byte loads per chunk and a bitwise OR
    // (buffer loads: SGPR descriptor + 32-bit lane offset + scalar piece offset -- no 64-bit per-lane pointers to keep alive)
    const __amdgpu_buffer_rsrc_t fq_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.FqH), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t fp_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.FpH), 0, 0x7fffffff, 0x00020000);
    auto load_inputs = [&](unsigned rt, unsigned ro, u4 (&raw)[4]) {
        unsigned hh = (unsigned)half;
        asm volatile("" : "+v"(hh));   // opaque: keeps hipcc from hoisting half * ld out of the tile loop into VGPRs it then spills
        const int vq = (int)((hh * (unsigned)a.ldFq + rt) * 16u), vp = (int)((hh * (unsigned)a.ldFp + ro) * 16u);
        raw[0] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(fq_rs, vq, 0, 0));
        raw[1] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(fp_rs, vp, 0, 0));
        raw[2] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(fq_rs, vq, a.ldFq * 32, 0));
        raw[3] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(fp_rs, vp, a.ldFp * 32, 0));
    };
    // skip-connection networks: the same encoded input at the slots of the concatenated columns.  The host packs the
    // layer behind a concatenation with those columns LAST (virtual columns 256 - 3d .. 255 = k-chunks 14 and 15, whatever
    // the width of the layer in front: the K order of a dot product is free), so the concatenation is a bitwise OR into
    // act[14], act[15] -- zeros there, the padded units of the narrower layer -- and one table serves every level
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t sq_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(SKIP ? a.FqS : a.FqH), 0, 0x7fffffff, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t sp_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(SKIP ? a.FpS : a.FpH), 0, 0x7fffffff, 0x00020000);
    auto load_skip = [&](unsigned rt, unsigned ro, u4 (&sk)[2]) {
        unsigned hh = (unsigned)half;
        asm volatile("" : "+v"(hh));
        const int vq = (int)((hh * (unsigned)a.ldFq + rt) * 16u), vp = (int)((hh * (unsigned)a.ldFp + ro) * 16u);
        sk[0] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(sq_rs, vq, 0, 0)) |
                __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(sp_rs, vp, 0, 0));
        sk[1] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(sq_rs, vq, a.ldFq * 32, 0)) |
                __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(sp_rs, vp, a.ldFp * 32, 0));
    };
    unsigned row_o[SC_RB];
    u4 raw[SC_RB][4];
    [[maybe_unused]] u4 skn[SC_RB][2];   // SKIP: the next tile's concatenation operands, fetched with its inputs
    float rad[SC_RB];
#pragma unroll
    for (int rb = 0; rb < SC_RB; ++rb) {
        unsigned rt;
        tile_row(a, p0, p1, 0, wave, rb, b, rt, row_o[rb]);
        load_inputs(rt, row_o[rb], raw[rb]);
        if constexpr (SKIP) load_skip(rt, row_o[rb], skn[rb]);
        rad[rb] = a.radius[row_o[rb]];
    }

    const unsigned char* ring_lane = ring + lane * 16;
    // accumulators start at the bias.  C layout: register r = 4j + i <-> output row 32 fb + 8 j + 4 half + i
    auto read_bias = [&](int layer, int fb) {
        f32x16 r;
        const float* bl = biasL + layer * OMDS_WIDTH + 32 * fb + 4 * half;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 bv = *reinterpret_cast<const float4*>(bl + 8 * j);
            r[4 * j] = bv.x; r[4 * j + 1] = bv.y; r[4 * j + 2] = bv.z; r[4 * j + 3] = bv.w;
        }
        return r;
    };
    auto to_act = [&](const f32x16& acc, h8& lo, h8& hi) {   // registers 0-7 are the slots of chunk 2 fb of the next layer, 8-15 of 2 fb + 1
        const h2 q0 = act_pk<ACT>(acc[0], acc[1]), q1 = act_pk<ACT>(acc[2], acc[3]), q2 = act_pk<ACT>(acc[4], acc[5]), q3 = act_pk<ACT>(acc[6], acc[7]);
        const h2 q4 = act_pk<ACT>(acc[8], acc[9]), q5 = act_pk<ACT>(acc[10], acc[11]), q6 = act_pk<ACT>(acc[12], acc[13]), q7 = act_pk<ACT>(acc[14], acc[15]);
        lo = h8{q0[0], q0[1], q1[0], q1[1], q2[0], q2[1], q3[0], q3[1]};
        hi = h8{q4[0], q4[1], q5[0], q5[1], q6[0], q6[1], q7[0], q7[1]};
    };

    // ---- the slice pipeline.  Global step sigma = it * S + s multiplies slice s (ring slot sigma % RING) in 4 groups of 4
    //      fragments; the A fragments of the next group are read while the current group's MFMAs issue.  In the middle of a
    //      step the wave waits for ITS pieces of the next slice (issued DIST-1 steps earlier) and meets the others at the
    //      barrier: after it that slice is complete and every wave has finished reading the previous one, so the slice DIST
    //      steps ahead may overwrite a slot last read RING - DIST >= 1 steps ago.  Slices keep being issued past the last
    //      tile (into free slots, never read), so the counted wait is one constant.
    wait_vm_barrier(SC_PW * (SC_DIST - 1));   // slice 0 landed (the later ones may still be in flight)
    AGroup cur = read_group(ring_lane, 0);
    int sigma0 = 0;
    for (int it = 0; it < my_tiles; ++it, sigma0 += S) {
        SC_TL(0);
        auto sync_and_issue = [&](int s) {
#ifdef OMDS_SC_EXPERIMENT   // timing experiments only (variant builds): OMDS_SCREEN_DBG 4 = no wait + barrier, 1 = no weight streaming
            if (!(OMDS_DBG(a.dbg) & 4)) wait_vm_barrier(SC_PW * (SC_DIST - 2));
            if (!(OMDS_DBG(a.dbg) & 1)) issue((s + SC_DIST) % S, (sigma0 + s + SC_DIST) & (SC_RING - 1));
#else
            wait_vm_barrier(SC_PW * (SC_DIST - 2));
            issue((s + SC_DIST) % S, (sigma0 + s + SC_DIST) & (SC_RING - 1));
#endif
        };
        auto slot_ptr = [&](int s) { return ring_lane + ((sigma0 + s) & (SC_RING - 1)) * SC_SLICE; };
#ifdef OMDS_SC_EXPERIMENT   // OMDS_SCREEN_DBG 8: the partial last tile of a chunk is not multiplied at all (what its round costs)
        if ((OMDS_DBG(a.dbg) & 8) && it == my_tiles - 1 && (my_pairs % SC_ROWS) != 0) {
            for (int s = 0; s < S; ++s) sync_and_issue(s);
            continue;
        }
#endif
        if (it * SC_ROWS + wave * SC_RB * 32 >= my_pairs) {
            // all pairs of this wave lie past the end of the chunk (the partial last tile of a chunk of whole rollouts): it keeps
            // the ring moving -- its DMA pieces, the barriers -- and issues no MFMA.  The matrix pipe is power-limited under this
            // kernel: multiplying clamped duplicates would slow the waves that have real pairs (measured: 99 -> 108 us at N = 1024)
            for (int s = 0; s < S; ++s) sync_and_issue(s);
            continue;
        }
        h8 in[SC_RB][2];
        [[maybe_unused]] h8 sk[SC_RB][2];
#pragma unroll
        for (int rb = 0; rb < SC_RB; ++rb) {
            const u4 i0 = raw[rb][0] | raw[rb][1], i1 = raw[rb][2] | raw[rb][3];
            in[rb][0] = __builtin_bit_cast(h8, i0);
            in[rb][1] = __builtin_bit_cast(h8, i1);
            if constexpr (SKIP) { sk[rb][0] = __builtin_bit_cast(h8, skn[rb][0]); sk[rb][1] = __builtin_bit_cast(h8, skn[rb][1]); }
        }
        h8 act[SC_RB][16];
        // ReLU networks: bit cc = some pair of this wave has a non-zero activation in k-chunk cc of the layer about to be consumed.
        // A chunk whose 16 units are zero for all 32 pairs contributes exactly nothing to any output: its MFMAs are not issued.
        // (The host orders the hidden units of the screening pack by how often they fire, so that the units a trained network
        //  never uses -- a third of the shipped one's -- fill whole chunks: omds_internal.h, screen reorder.)
        unsigned alive = 0xffffu;
        auto find_alive = [&]() {
            if constexpr (ACT == OMDS_ACT_RELU && !SKIP) {
                unsigned mk = (1u << SC_TEST0) - 1u;   // the first chunks hold the units that fire most: multiplied without a test
#pragma unroll
                for (int cc = SC_TEST0; cc < 16; ++cc) {
                    unsigned nz = 0;
#pragma unroll
                    for (int rb = 0; rb < SC_RB; ++rb) {
                        const u4 v = __builtin_bit_cast(u4, act[rb][cc]);
                        nz |= v.x | v.y | v.z | v.w;
                    }
                    mk |= (__builtin_amdgcn_ballot_w64(nz != 0u) != 0ull ? 1u : 0u) << cc;
                }
                alive = mk;
            }
        };
        auto concat = [&](int level) {   // wave-uniform test; the OR touches 8 registers
            if constexpr (SKIP) {
                if ((a.skip_mask >> level) & 1u) {
#pragma unroll
                    for (int rb = 0; rb < SC_RB; ++rb) {
                        act[rb][14] = __builtin_bit_cast(h8, __builtin_bit_cast(u4, act[rb][14]) | __builtin_bit_cast(u4, sk[rb][0]));
                        act[rb][15] = __builtin_bit_cast(h8, __builtin_bit_cast(u4, act[rb][15]) | __builtin_bit_cast(u4, sk[rb][1]));
                    }
                }
            }
        };
        // ---- step 0: layer 1 on the matrix pipe.  Slice 0 = W1 as 8 row blocks x 2 k-chunks (fragment 2 fb + cc)
        {
            const unsigned char* sl = slot_ptr(0);
            const unsigned char* sl_next = slot_ptr(1);
            // group g = row blocks 2g, 2g+1 (two MFMAs each and pair block); their activation + conversion runs one group later,
            // under the next group's MFMAs; the scheduling barriers keep hipcc from batching all MFMAs first
            f32x16 pa[SC_RB], pb[SC_RB];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g == 1) sync_and_issue(0);
                const AGroup pre = (g < 3) ? read_group(sl, g + 1) : read_group(sl_next, 0);
                const f32x16 ba = read_bias(0, 2 * g), bb = read_bias(0, 2 * g + 1);
                f32x16 ca[SC_RB], cb[SC_RB];
#pragma unroll
                for (int rb = 0; rb < SC_RB; ++rb) {
                    ca[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.f[0], in[rb][0], ba, 0, 0, 0);
                    cb[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.f[2], in[rb][0], bb, 0, 0, 0);
                }
#pragma unroll
                for (int rb = 0; rb < SC_RB; ++rb) {
                    ca[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.f[1], in[rb][1], ca[rb], 0, 0, 0);
                    cb[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.f[3], in[rb][1], cb[rb], 0, 0, 0);
                }
                if (g > 0) {
#pragma unroll
                    for (int rb = 0; rb < SC_RB; ++rb) {
                        to_act(pa[rb], act[rb][4 * g - 4], act[rb][4 * g - 3]);
                        to_act(pb[rb], act[rb][4 * g - 2], act[rb][4 * g - 1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < SC_RB; ++rb) { pa[rb] = ca[rb]; pb[rb] = cb[rb]; }
                cur = pre;
            }
#pragma unroll
            for (int rb = 0; rb < SC_RB; ++rb) {
                to_act(pa[rb], act[rb][12], act[rb][13]);
                to_act(pb[rb], act[rb][14], act[rb][15]);
            }
            concat(0);
            find_alive();
        }
        SC_TL(1);
        // ---- steps 1 .. S-1: hidden->hidden layers and the last layer
        h8 nxt[SC_RB][16];
        float dmin[SC_RB];
        f32x16 acc[SC_RB];
        {
            const f32x16 b0 = read_bias(1, 0);
#pragma unroll
            for (int rb = 0; rb < SC_RB; ++rb) { acc[rb] = b0; dmin[rb] = __builtin_inff(); }
        }
        unsigned nrow_o[SC_RB];
#pragma unroll
        for (int rb = 0; rb < SC_RB; ++rb) nrow_o[rb] = row_o[rb];
#pragma unroll
        for (int s = 1; s < S; ++s) {
            const int fb = (s - 1) & 7;                        // output row block of hidden->hidden layer (s - 1) >> 3 (bias row + 1)
            const bool last = s == S - 1;
            const unsigned char* sl = slot_ptr(s);
            const unsigned char* sl_next = slot_ptr(s + 1);
            f32x16 acc_next = acc[0];                          // the next step's start values (the bias: the same for every pair block)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g == 1) sync_and_issue(s);
                const AGroup pre = (g < 3) ? read_group(sl, g + 1) : read_group(sl_next, 0);   // (reading only the fragments that will be multiplied: 96 -> 103 us)
                if (g == 2 && !last) acc_next = (s + 1 == S - 1) ? read_bias(NHH + 1, 0) : read_bias(((s) >> 3) + 1, s & 7);
                if (s == S - 2 && g == 2) {   // the next tile's inputs, fetched under the last two steps of this one
#pragma unroll
                    for (int rb = 0; rb < SC_RB; ++rb) {
                        unsigned nt;
                        tile_row(a, p0, p1, it + 1, wave, rb, b, nt, nrow_o[rb]);
                        load_inputs(nt, nrow_o[rb], raw[rb]);
                        if constexpr (SKIP) load_skip(nt, nrow_o[rb], skn[rb]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (4 * g + i < SC_TEST0 || ((alive >> (4 * g + i)) & 1u)) {
#pragma unroll
                        for (int rb = 0; rb < SC_RB; ++rb)
                            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.f[i], act[rb][4 * g + i], acc[rb], 0, 0, 0);
                    }
                }
                // issue order inside the group: one fragment read of the NEXT group ahead of each fragment's MFMAs (the
                // group's MFMAs = 128+ cycles of lead for the LDS latency; group 2 also carries the four bias reads of the
                // next step); everything else (the previous step's epilogue VALU) fills in behind
                if (g == 2 && !last) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, SC_RB, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, SC_RB, 0);
                    }
                }
                cur = pre;
            }
            if (!last) {
#pragma unroll
                for (int rb = 0; rb < SC_RB; ++rb) {
                    to_act(acc[rb], nxt[rb][2 * fb], nxt[rb][2 * fb + 1]);
                    if (fb == 7) {
#pragma unroll
                        for (int cc = 0; cc < 16; ++cc) act[rb][cc] = nxt[rb][cc];
                    }
                }
                if (fb == 7) { concat(((s - 1) >> 3) + 1); find_alive(); }
            } else {
                // last layer: the links sit on the A rows (rows >= C are zero); min over the valid, un-ignored links
#pragma unroll
                for (int rb = 0; rb < SC_RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 8; ++r) {   // link = 4 half + (r & 3) + 8 (r >> 2); links >= 16 do not exist (OMDS_CPAD)
                        const int link = 4 * half + (r & 3) + 8 * (r >> 2);
                        float v = acc[rb][r] / a.out_div - rad[rb];
                        v = (link >= a.C) ? __builtin_inff() : (((a.ignored >> link) & 1u) ? 1e6f : v);
                        dmin[rb] = fminf(dmin[rb], v);
                    }
            }
#pragma unroll
            for (int rb = 0; rb < SC_RB; ++rb) acc[rb] = acc_next;
            if (s == 8) SC_TL(2);
        }
        SC_TL(3);
        // links 0-3, 8-11 sit in lane-half 0, the others in half 1: both min into the pair's LDS slot (ds_min_f32; a cross-lane
        // exchange would keep a lane-index VGPR alive across the whole tile loop)
#pragma unroll
        for (int rb = 0; rb < SC_RB; ++rb) {
            __hip_atomic_fetch_min(&resL[it * SC_ROWS + (wave * SC_RB + rb) * 32 + b], dmin[rb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            row_o[rb] = nrow_o[rb];
            rad[rb] = a.radius[row_o[rb]];
        }
    }
    if (a.tl && threadIdx.x == 0) a.tl[(size_t)blockIdx.x * 8 + 4] = __builtin_readcyclecounter();
    // ---- drain the ring (pieces issued past the last tile still target this workgroup's LDS), then flush the results
    wait_vm_barrier(0);
    __syncthreads();
    if (a.sel) {
        // the chunk is whole rollouts: resL[rl * O .. +O) are the screening values of rollout t0 + rl
        SelectShared& sh = *reinterpret_cast<SelectShared*>(ring);   // the ring is drained and idle
        const SelectSink sel = *a.sel;
        const int O = a.O;
        const int t0 = (int)a.odiv.div((unsigned)p0);
        const int n_roll = my_pairs / O;
        for (int r0 = 0; r0 < n_roll; r0 += SEL_WAVES) {
            const int rl = r0 + wave;
            const bool live = rl < n_roll;
            const int rr = live ? rl : n_roll - 1;
            select_rollout(sel, sh, (const __attribute__((address_space(3))) float*)resL + rr * O, t0 + rr, live, O, lane, wave);
        }
    } else {
        for (int i = tid; i < my_pairs; i += SC_NT) a.Dmin[p0 + i] = resL[i];
    }
    if (a.tl && threadIdx.x == 0) a.tl[(size_t)blockIdx.x * 8 + 6] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// k_select: one wave per rollout (select_rollout above), for the steps whose k_screen writes the matrix: rows too long for a
// workgroup's LDS, and the matrix route of a tanh network whose derivative hand-over buffer could not be allocated (k_exact mode 3
// writes the exact values into the matrix).
// ------------------------------------------------------------------------------------------------
struct SelectArgs {
    const float* Dmin;    // [N][O] screening values
    int N, O;
    SelectSink sel;
};
__global__ __launch_bounds__(SEL_WAVES * 64) void k_select(SelectArgs a) {
    __shared__ SelectShared sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t_raw = blockIdx.x * SEL_WAVES + wave;
    const bool live = t_raw < a.N;                       // waves past the last rollout take part in the barriers with 0 entries
    const int t = live ? t_raw : a.N - 1;
    select_rollout(a.sel, sh, a.Dmin + (size_t)t * a.O, t, live, a.O, lane, wave);
}

// ------------------------------------------------------------------------------------------------
// k_exact: fp32 pass-1 tiles over the listed rows, 16 rows per tile (v_mfma_f32_16x16x4 fed in the k order of the 32-row
// kernels: the same bits per row, mlp_device.h), up to three workgroups resident per CU (78 registers).  A tile is a chain of
// dependent GEMMs on one CU, so what counts is (a) the granularity -- 9.2 candidates per rollout at N = 1024 are 590 tiles of
// 16 rows for 768 slots, where 32-row tiles were 295 for 256 CUs and the 39 left over doubled the launch time -- and (b)
// co-residency: three tiles on a CU fill each other's barrier / epilogue / last-layer gaps.  Measured at 9.9 candidates per
// rollout: 52.9 us, against 59.6 (32-row tiles, two resident), 64.1 (an equal share of 32 + 16 rows per CU, one after the
// other) and 58.0 (16-row tiles, two resident).
// ------------------------------------------------------------------------------------------------
template <int ACT, int MODE>
__global__ __launch_bounds__(512, 6) void k_exact(MlpDev m, const float* __restrict__ Fq, const float* __restrict__ Fp,
                                                  const float* __restrict__ radius, int O, uint32_t ignored,
                                                  float* __restrict__ Dmin, OmdsDivisor odiv, const int* __restrict__ rowlist,
                                                  const int* __restrict__ total, unsigned* __restrict__ maxerr_bits, ExactOut ex) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = *total;
    for (int blk = blockIdx.x; blk * 16 < n; blk += gridDim.x) {
        pass1_tile<16, 1, 1, ACT, MODE>(m, smem, Fq, Fp, radius, O, n, ignored, Dmin, (long long)blk * 16, odiv, rowlist, maxerr_bits, &ex);
        __syncthreads();   // the tile buffer is reused by the next tile
    }
}

// ------------------------------------------------------------------------------------------------
// k_audit: the audit sample of a whole propagate (k_select recorded (row, screening value) of a pseudo-random subset of the
// pairs that were NOT candidates) evaluated in fp32 with the pass-1 tile code -- 64-row tiles, the throughput shape of
// k_pass1, because nothing waits for this launch but the end of the propagate -- against the layer-1 table of ALL horizon
// steps' states ([H*N][256], rebuilt from the stored rollouts by k_rollout_layer1, the arithmetic of the step itself).
// Output: max (Da - D) over the sample, the one-sided error the selection rule bounds by eps (maxerr_bits[2]).
// ------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(512) void k_audit(MlpDev m, const float* __restrict__ FqAll, const float* __restrict__ Fp,
                                               const float* __restrict__ radius, int O, uint32_t ignored, OmdsDivisor odiv,
                                               const int* __restrict__ rows, const int* __restrict__ total, int cap,
                                               unsigned* __restrict__ maxerr_bits, ExactOut ex) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = min(*total, cap);
    // whole rounds of 64-row tiles over the grid; a last round that would fill at most half the grid's row slots runs on 32-row
    // tiles, half as long (72 k rows on 512 workgroups are 2.2 rounds of 64 rows: 2 + a half instead of 3)
    const int G = (int)gridDim.x;
    const int full = n / (64 * G);                  // rounds in which every workgroup has a full tile
    const int rest0 = full * 64 * G, rest = n - rest0;
    for (int r = 0; r < full; ++r) {
        pass1_tile<64, 2, 1, ACT, 4>(m, smem, FqAll, Fp, radius, O, n, ignored, nullptr, ((long long)r * G + blockIdx.x) * 64, odiv, rows, maxerr_bits, &ex);
        __syncthreads();
    }
    if (rest > 32 * G) {
        if ((long long)blockIdx.x * 64 < rest)
            pass1_tile<64, 2, 1, ACT, 4>(m, smem, FqAll, Fp, radius, O, n, ignored, nullptr, rest0 + (long long)blockIdx.x * 64, odiv, rows, maxerr_bits, &ex);
    } else if ((long long)blockIdx.x * 32 < rest) {
        pass1_tile<32, 1, 1, ACT, 4>(m, smem, FqAll, Fp, radius, O, n, ignored, nullptr, rest0 + (long long)blockIdx.x * 32, odiv, rows, maxerr_bits, &ex);
    }
}

void omds_launch_audit(hipStream_t s, const MlpDev& m, const float* FqAll, const float* Fp, const float* radius, int O,
                       uint32_t ignored, const int* rows, const float* da, const int* total, int cap, unsigned* maxerr_bits) {
    if (cap <= 0) return;
    const size_t lds = (size_t)64 * LDH * 4 + 64 * 4 + 64 * 4;
    static std::atomic<uint64_t> configured{0};
    if (omds_first_use_on_device(configured)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_audit<OMDS_ACT_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_audit<OMDS_ACT_TANH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    ExactOut ex{};
    ex.Da = da;
    // the list length is only known on the device: a grid of two resident 64-row workgroups per CU strides over it
    const unsigned grid = (unsigned)std::min<long long>(((long long)cap + 63) / 64, 512);
    const OmdsDivisor od = OmdsDivisor::make((unsigned)O);
    if (m.act == OMDS_ACT_RELU)
        hipLaunchKernelGGL((k_audit<OMDS_ACT_RELU>), dim3(grid), dim3(512), lds, s, m, FqAll, Fp, radius, O, ignored, od, rows, total, cap, maxerr_bits, ex);
    else
        hipLaunchKernelGGL((k_audit<OMDS_ACT_TANH>), dim3(grid), dim3(512), lds, s, m, FqAll, Fp, radius, O, ignored, od, rows, total, cap, maxerr_bits, ex);
}

// ------------------------------------------------------------------------------------------------
// Calibration of the screening bound on the device (capi.hip: calibrate_screen): B states -- the even ones uniform inside the
// joint box, the odd ones drawn from the rollouts of the last propagate (where the next rollouts will live), or scattered
// around the start state while there are none yet -- and the largest |screening value - fp32 value| over their B x O pairs.
// ------------------------------------------------------------------------------------------------
struct CalibArgs {
    float* qT;             // [n][B] out
    int B, n;
    float lo[OMDS_MAX_DOF], hi[OMDS_MAX_DOF], center[OMDS_MAX_DOF];
    const float* trajT;    // [H][n][N] rollouts of the last propagate, or nullptr
    int N, H;
    unsigned seed;
};
__global__ __launch_bounds__(256) void k_calib_states(CalibArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.B) return;
    auto uni = [&](unsigned salt) { return ((omds_audit_hash((unsigned)t * 0x9E3779B1u + salt + a.seed) >> 8) + 0.5f) * (1.f / 16777216.f); };
    if ((t & 1) && a.trajT) {
        const unsigned r = omds_audit_hash((unsigned)t + 0x51ED270Bu * a.seed);
        const int tt = (int)(r % (unsigned)a.N), hh = (int)((r >> 16) % (unsigned)a.H);
        for (int j = 0; j < a.n; ++j) a.qT[(size_t)j * a.B + t] = a.trajT[((size_t)hh * a.n + j) * a.N + tt];
        return;
    }
    for (int j = 0; j < a.n; ++j) {
        float v;
        if (t & 1) {   // Box-Muller scatter (sigma 0.6 rad) around the start state, clamped to the box
            const float g = sqrtf(-2.f * logf(uni(2 * j + 1))) * cosf(6.2831853f * uni(2 * j + 2));
            v = fminf(a.hi[j], fmaxf(a.lo[j], a.center[j] + 0.6f * g));
        } else {
            v = a.lo[j] + (a.hi[j] - a.lo[j]) * uni(2 * j + 1);
        }
        a.qT[(size_t)j * a.B + t] = v;
    }
}
void omds_launch_calib_states(hipStream_t s, float* qT, int B, int n, const float* lo, const float* hi, const float* center,
                              const float* trajT, int N, int H, unsigned seed) {
    CalibArgs a;
    a.qT = qT; a.B = B; a.n = n; a.trajT = trajT; a.N = N; a.H = H; a.seed = seed;
    for (int j = 0; j < OMDS_MAX_DOF; ++j) { a.lo[j] = j < n ? lo[j] : 0.f; a.hi[j] = j < n ? hi[j] : 0.f; a.center[j] = j < n ? center[j] : 0.f; }
    hipLaunchKernelGGL(k_calib_states, dim3((B + 255) / 256), dim3(256), 0, s, a);
}
// *out_bits = max(*out_bits, max_i |x[i] - y[i]|) as float bits; a non-finite difference counts as +inf
__global__ __launch_bounds__(256) void k_max_abs_diff(const float* __restrict__ x, const float* __restrict__ y, long long n, unsigned* out_bits) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float e = fabsf(x[i] - y[i]);
        if (!(e <= m)) m = (e < __builtin_inff()) ? e : __builtin_inff();
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out_bits, __builtin_bit_cast(unsigned, m));
}
void omds_launch_max_abs_diff(hipStream_t s, const float* x, const float* y, long long n, unsigned* out_bits) {
    if (n <= 0) return;
    const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(k_max_abs_diff, dim3(grid), dim3(256), 0, s, x, y, n, out_bits);
}

// Sweep statistics (omds.h, omds_screen_sweep_hist): one step's N x O screening values Da beside its N x O fp32 values D and the
// tau of every rollout (range[4 t + 2], what the selection of that step used).  x = Da - D over the pairs that are NOT candidates
// -- the population the selection rule's assumption "x <= eps" is about -- goes into histograms; max |x| over ALL pairs into
// *maxabs_bits (a non-finite difference counts as +inf) like k_max_abs_diff.  Counters are 64-bit and accumulate across launches.
__global__ __launch_bounds__(256) void k_sweep_hist(const float* __restrict__ D, const float* __restrict__ Da, const int* __restrict__ range,
                                                    int N, int O, OmdsDivisor od, float eps, unsigned long long* __restrict__ hist, unsigned* maxabs_bits) {
    __shared__ unsigned bins[OMDS_HIST_LOG_BINS * 2 + OMDS_HIST_RATIO_BINS];
    __shared__ unsigned cnt[8];
    for (int i = threadIdx.x; i < OMDS_HIST_LOG_BINS * 2 + OMDS_HIST_RATIO_BINS; i += blockDim.x) bins[i] = 0;
    if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
    __syncthreads();
    const long long total = (long long)N * O;
    float mabs = 0.f, mpos = 0.f;
    unsigned n_all = 0, n_non = 0, n_half = 0, n_eps = 0, n_bad = 0;
    const float inv_eps = eps > 0.f ? 1.f / eps : 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const float d = D[i], da = Da[i];
        const int t = (int)od.div((unsigned)i);
        const float tau = __builtin_bit_cast(float, range[4 * t + 2]);
        const float x = da - d, e = fabsf(x);
        ++n_all;
        if (!(e <= mabs)) mabs = (e < __builtin_inff()) ? e : __builtin_inff();
        const bool cand = !(da > tau) || !(da < __builtin_inff());   // select_rollout's is_cand
        if (cand) continue;
        ++n_non;
        if (!(e < __builtin_inff())) { ++n_bad; continue; }          // a finite Da beside a non-finite D: cannot happen with finite weights
        if (x > mpos) mpos = x;
        if (x > 0.5f * eps) ++n_half;
        if (x > eps) ++n_eps;
        // log2 bins: bin b holds 2^(b - OMDS_HIST_LOG_BINS) <= |x| < 2^(b + 1 - OMDS_HIST_LOG_BINS); bin 0 also everything smaller
        // (zero included), the last bin everything >= 1/2
        int b = (e > 0.f ? ilogbf(e) : -1000) + OMDS_HIST_LOG_BINS;
        b = b < 0 ? 0 : (b > OMDS_HIST_LOG_BINS - 1 ? OMDS_HIST_LOG_BINS - 1 : b);
        atomicAdd(&bins[(x < 0.f ? OMDS_HIST_LOG_BINS : 0) + b], 1u);
        if (x > 0.f) {   // x / eps in OMDS_HIST_RATIO_BINS linear bins over [0, 1); the last bin also everything >= 1
            int r = (int)(x * inv_eps * OMDS_HIST_RATIO_BINS);
            r = r > OMDS_HIST_RATIO_BINS - 1 ? OMDS_HIST_RATIO_BINS - 1 : r;
            atomicAdd(&bins[2 * OMDS_HIST_LOG_BINS + r], 1u);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { mabs = fmaxf(mabs, __shfl_xor(mabs, off)); mpos = fmaxf(mpos, __shfl_xor(mpos, off)); }
    atomicAdd(&cnt[0], n_all); atomicAdd(&cnt[1], n_non); atomicAdd(&cnt[2], n_half); atomicAdd(&cnt[3], n_eps); atomicAdd(&cnt[4], n_bad);
    if ((threadIdx.x & 63) == 0) {
        if (mabs > 0.f) atomicMax(maxabs_bits, __builtin_bit_cast(unsigned, mabs));
        if (mpos > 0.f) atomicMax(&hist[OMDS_HIST_MAX_POS], (unsigned long long)__builtin_bit_cast(unsigned, mpos));
        if (mabs > 0.f) atomicMax(&hist[OMDS_HIST_MAX_ABS], (unsigned long long)__builtin_bit_cast(unsigned, mabs));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&hist[OMDS_HIST_PAIRS], (unsigned long long)cnt[0]);
        atomicAdd(&hist[OMDS_HIST_NONCAND], (unsigned long long)cnt[1]);
        atomicAdd(&hist[OMDS_HIST_ABOVE_HALF], (unsigned long long)cnt[2]);
        atomicAdd(&hist[OMDS_HIST_ABOVE_EPS], (unsigned long long)cnt[3]);
        atomicAdd(&hist[OMDS_HIST_NONFINITE], (unsigned long long)cnt[4]);
    }
    for (int i = threadIdx.x; i < OMDS_HIST_LOG_BINS * 2 + OMDS_HIST_RATIO_BINS; i += blockDim.x)
        if (bins[i]) atomicAdd(&hist[OMDS_HIST_BINS0 + i], (unsigned long long)bins[i]);
}
void omds_launch_sweep_hist(hipStream_t s, const float* D, const float* Da, const int* range, int N, int O, float eps,
                            unsigned long long* hist, unsigned* maxabs_bits) {
    const long long n = (long long)N * O;
    if (n <= 0) return;
    const unsigned grid = (unsigned)std::min<long long>((n + 2047) / 2048, 512);   // few workgroups: each ends with ~200 64-bit atomics
    hipLaunchKernelGGL(k_sweep_hist, dim3(grid), dim3(256), 0, s, D, Da, range, N, O, OmdsDivisor::make((unsigned)O), eps, hist, maxabs_bits);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
size_t omds_screen_lds_bytes(int nhh) { return (size_t)SC_RING * SC_SLICE + (size_t)(nhh + 2) * OMDS_WIDTH * 4; }

int omds_cu_count() {
    static std::atomic<int> ncu_of[64];   // CUs per device, asked once
    int dev = 0;
    (void)hipGetDevice(&dev);
    int ncu = ncu_of[dev & 63].load();
    if (ncu == 0) {
        ncu = 256;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu <= 0) ncu = 256;
        ncu_of[dev & 63].store(ncu);
    }
    return ncu;
}

// the flush phase can select when a rollout's O screening values fit a workgroup's result buffer
bool omds_screen_can_select(int O) { return O <= SC_MAX_TILES * SC_ROWS; }

template <int NHH, int ACT, bool SKIP>
static void launch_screen_t(hipStream_t s, dim3 grid, size_t lds, size_t lds_max, const ScreenArgs& a) {
    static std::atomic<uint64_t> configured{0};
    if (omds_first_use_on_device(configured))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_screen<NHH, ACT, SKIP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    hipLaunchKernelGGL((k_screen<NHH, ACT, SKIP>), grid, dim3(SC_NT), lds, s, a);
}
template <int NHH>
static void launch_screen_n(hipStream_t s, dim3 grid, size_t lds, size_t lds_max, const ScreenArgs& a, int act, bool skip) {
    if (act == OMDS_ACT_RELU) { if (skip) launch_screen_t<NHH, OMDS_ACT_RELU, true>(s, grid, lds, lds_max, a); else launch_screen_t<NHH, OMDS_ACT_RELU, false>(s, grid, lds, lds_max, a); }
    else { if (skip) launch_screen_t<NHH, OMDS_ACT_TANH, true>(s, grid, lds, lds_max, a); else launch_screen_t<NHH, OMDS_ACT_TANH, false>(s, grid, lds, lds_max, a); }
}

void omds_launch_screen(hipStream_t s, const ScreenDev& sd, const MlpDev& m, const uint16_t* FqH, int ldFq, const uint16_t* FpH,
                        int ldFp, const float* radius, int O, int B, uint32_t ignored, float* Dmin, const SelectSink* sel) {
    // sel: DEVICE pointer to this step's sink (capi.hip uploads the sinks of all horizon steps before the loop)
    const long long total = (long long)B * O;
    if (total <= 0) return;
    ScreenArgs a;
    a.Wh = reinterpret_cast<const _Float16*>(sd.Wh);
    a.bias = sd.bias;
    a.FqH = reinterpret_cast<const _Float16*>(FqH); a.FpH = reinterpret_cast<const _Float16*>(FpH);
    a.FqS = reinterpret_cast<const _Float16*>(m.scrQ); a.FpS = reinterpret_cast<const _Float16*>(m.scrP);
    a.ldFq = ldFq; a.ldFp = ldFp;
    a.radius = radius; a.Dmin = Dmin; a.B = B;
    a.total_rows = total; a.O = O; a.ignored = ignored; a.odiv = OmdsDivisor::make((unsigned)O);
    a.nhh = m.nhh; a.C = m.C; a.out_div = m.out_div;
    a.skip_mask = m.skip_mask;
    a.sel = nullptr;
    static const int dbg = OMDS_EXP_ENV("OMDS_SCREEN_DBG", 0);
    a.dbg = dbg;
    // persistent: one workgroup per CU (256 on MI355X; more only when a workgroup's result buffer would overflow its LDS),
    // each with a contiguous chunk of the pair space
    int ncu = omds_cu_count();
    static const int cu_cap = OMDS_EXP_ENV("OMDS_SCREEN_CUS", 0);   // experiment builds: persistent workgroups on a subset of the CUs
    if (cu_cap > 0) ncu = std::min(ncu, cu_cap);
    long long gl;
    int tiles_per_wg;
    if (sel && omds_screen_can_select(O)) {
        // whole rollouts per workgroup: the flush phase selects (no matrix, no k_select)
        const long long Rmax = std::max<long long>(1, (long long)SC_MAX_TILES * SC_ROWS / O);   // rollouts whose values fit the result buffer
        gl = std::max<long long>(std::min<long long>(B, ncu), ((long long)B + Rmax - 1) / Rmax);
        a.unit = O;
        a.units_base = (int)(B / gl);
        a.units_rem = (int)(B % gl);
        tiles_per_wg = (int)(((long long)(a.units_base + (a.units_rem ? 1 : 0)) * O + SC_ROWS - 1) / SC_ROWS);
        a.sel = sel;
        a.Dmin = nullptr;
    } else {
        const long long ntiles = (total + SC_ROWS - 1) / SC_ROWS;
        gl = std::min<long long>(ntiles, ncu);
        gl = std::max<long long>(gl, (ntiles + SC_MAX_TILES - 1) / SC_MAX_TILES);
        tiles_per_wg = (int)((ntiles + gl - 1) / gl);
        a.unit = SC_ROWS;
        a.units_base = (int)(ntiles / gl);
        a.units_rem = (int)(ntiles % gl);
    }
    const dim3 grid((unsigned)gl);
    const size_t lds = omds_screen_lds_bytes(m.nhh) + (size_t)tiles_per_wg * SC_ROWS * 4;
    a.res_tiles = tiles_per_wg;
    const size_t lds_max = omds_screen_lds_bytes(4) + (size_t)SC_MAX_TILES * SC_ROWS * 4;
    // diagnostic timeline (OMDS_SCREEN_TL=1): phase stamps of every workgroup's FIRST tile, summarised on stderr
    static const int tl_on = OMDS_EXP_ENV("OMDS_SCREEN_TL", 0);
    static unsigned long long* tl_buf = nullptr;
    a.tl = nullptr;
    if (tl_on && grid.x <= 65536) {
        if (!tl_buf) (void)hipMalloc(&tl_buf, (size_t)65536 * 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(tl_buf, 0, (size_t)grid.x * 8 * sizeof(unsigned long long), s);
        a.tl = tl_buf;
    }
    const bool skip = m.skip_mask != 0;
    switch (m.nhh) {
        case 1: launch_screen_n<1>(s, grid, lds, lds_max, a, m.act, skip); break;
        case 2: launch_screen_n<2>(s, grid, lds, lds_max, a, m.act, skip); break;
        case 3: launch_screen_n<3>(s, grid, lds, lds_max, a, m.act, skip); break;
        case 4: launch_screen_n<4>(s, grid, lds, lds_max, a, m.act, skip); break;
        default: break;   // omds_screen_supported() keeps other depths on the fp32 path
    }
    if (a.tl) {
        std::vector<unsigned long long> h((size_t)grid.x * 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), tl_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double seg[3] = {0, 0, 0}, pro = 0, loop = 0, wall = 0;
        unsigned long long w0 = ~0ull, w1 = 0;
        for (unsigned w = 0; w < grid.x; ++w) {
            const unsigned long long* r = &h[(size_t)w * 8];
            for (int i = 0; i < 3; ++i) seg[i] += (double)(r[i + 1] - r[i]);
            pro += (double)(r[0] - r[7]);          // kernel start -> first tile
            loop += (double)(r[4] - r[0]);         // all tiles of the workgroup
            wall += (double)(r[6] - r[5]);         // 100 MHz wall clock, whole workgroup
            w0 = std::min(w0, r[5]); w1 = std::max(w1, r[6]);
        }
        fprintf(stderr, "[k_screen timeline] per workgroup: prologue %.0f cycles, tile loop %.0f cycles, lifetime %.2f us (first start -> last end "
                        "%.2f us): %.2f GHz over the loop if the rest of the lifetime ran at the same clock\n", pro / grid.x, loop / grid.x,
                wall / grid.x / 100.0, (double)(w1 - w0) / 100.0, (pro + loop) / grid.x / (wall / grid.x / 100.0) / 1000.0);
        fprintf(stderr, "[k_screen timeline] %u workgroups x %d tiles; first tile, mean cycles: layer-1 step %.0f, first hidden layer %.0f, "
                        "remaining layers %.0f\n", grid.x, tiles_per_wg, seg[0] / grid.x, seg[1] / grid.x, seg[2] / grid.x);
    }
}

bool omds_screen_supported(const MlpDev& m) { return (m.act == OMDS_ACT_RELU || m.act == OMDS_ACT_TANH) && m.nhh >= 1 && m.nhh <= 4; }

void omds_launch_select(hipStream_t s, const float* Dmin, int B, int O, const SelectSink& sel) {
    if (B <= 0) return;
    SelectArgs a;
    a.Dmin = Dmin; a.N = B; a.O = O; a.sel = sel;
    hipLaunchKernelGGL(k_select, dim3((B + SEL_WAVES - 1) / SEL_WAVES), dim3(SEL_WAVES * 64), 0, s, a);
}

// ReLU networks: mode 1 -- per entry the exact value, pass 2's distance / arg-min link and the ReLU masks, for k_tail_sel.
// tanh networks: mode 5 -- the same with 1 - h^2 of every hidden unit (ex.deriv: 1 KB per entry and layer) in place of the masks, so
// that k_tail_sel<..., tanh> runs the backward only (round 4; 3 KB per candidate row of HBM traffic against a second forward in
// the tail).  Without ex.deriv (a matrix-mode step: rows too long for the selecting flush AND no derivative buffer), mode 3: the
// exact value replaces the screening value in the matrix Dmin itself and k_tail takes its top-k from the matrix and runs its own
// forward.
void omds_launch_exact(hipStream_t s, const MlpDev& m, const float* Fq, const float* Fp, const float* radius, int O,
                       int B, uint32_t ignored, float* Dmin, const int* rowlist, const int* total, unsigned* maxerr_bits,
                       const ExactOut& ex) {
    if (B <= 0) return;
    const size_t lds = (size_t)16 * LDH * 4 + 16 * 4 + 16 * 4 + (size_t)16 * (m.nhh + 1) * 8 * 4;
    const int ncu = omds_cu_count();
    // three resident workgroups per CU; longer lists stride (the list length is only known on the device)
    const long long blocks_max = ((long long)B * O + 15) / 16;
    static const int res = std::max(1, OMDS_EXP_ENV("OMDS_EXACT_RESIDENT", 3));   // workgroups per CU the grid is sized for (experiment builds: another count)
    const unsigned grid = (unsigned)std::min<long long>(blocks_max, (long long)res * ncu);
    const OmdsDivisor od = OmdsDivisor::make((unsigned)O);
    if (m.act == OMDS_ACT_RELU)
        hipLaunchKernelGGL((k_exact<OMDS_ACT_RELU, 1>), dim3(grid), dim3(512), lds, s, m, Fq, Fp, radius, O, ignored, Dmin, od, rowlist, total, maxerr_bits, ex);
    else if (ex.deriv)   // the derivative hand-over: k_tail_sel runs the backward on what this launch leaves per entry
        hipLaunchKernelGGL((k_exact<OMDS_ACT_TANH, 5>), dim3(grid), dim3(512), lds, s, m, Fq, Fp, radius, O, ignored, Dmin, od, rowlist, total, maxerr_bits, ex);
    else
        hipLaunchKernelGGL((k_exact<OMDS_ACT_TANH, 3>), dim3(grid), dim3(512), lds, s, m, Fq, Fp, radius, O, ignored, Dmin, od, rowlist, total, maxerr_bits, ex);
}

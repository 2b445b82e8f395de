// Device-side building blocks shared by the distance-network kernels (mlp_kernels.hip) and the fused
// per-step tail kernel (tail_kernel.hip): fp32 MFMA GEMM core, MFMA C-layout helper, pass-2 body.
#pragma once
#include "omds_internal.h"
#include "trig_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDH OMDS_LDH

// Diagnostic build (-DOMDS_TAIL_TL, tools/tail_timeline.sh): thread 0 of every k_tail_sel workgroup stamps the shader clock
// at the phase boundaries; a one-thread kernel prints the table after the launch chosen by OMDS_TAIL_TL_STEP.
#ifdef OMDS_TAIL_TL
static __device__ unsigned long long g_tail_tl[1024][20];
#define OMDS_TL_STAMP(i)                                                                                   \
    do {                                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_tail_tl[blockIdx.x][i] = (i) == 0 || (i) == 19 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define OMDS_TL_STAMP(i)
#endif

// hidden activation: ReLU (all shipped reference networks) or tanh (MATLAB-prototype style nets)
__device__ __forceinline__ float actf(float z, int act) { return act == OMDS_ACT_RELU ? fmaxf(z, 0.f) : tanhf(z); }


// ------------------------------------------------------------------------------------------------
// [MR*32 x 256] . [256 x NR*32] on v_mfma_f32_32x32x2_f32
//   A (activations) from LDS: lane l reads H[row = l&31 (+32 per row block)][8c + 4(l>>5) .. +3]
//   B (weights) from global, packed so that lane l's float4 = W[32 cb + (l&31)][8c + 4(l>>5) .. +3]
//   MFMA step m of chunk c contracts k = 8c + m (lanes 0-31) and k = 8c + 4 + m (lanes 32-63).
// ------------------------------------------------------------------------------------------------
template <int MR, int NR>
__device__ __forceinline__ void mfma_chunk(const float4 (&a)[MR], const float4 (&w)[NR], f32x16 (&acc)[MR][NR]) {
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, w[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, w[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, w[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, w[j].w, acc[i][j], 0, 0, 0);
}

// Weight fragments come through a buffer descriptor (SGPR base + per-lane byte offset + SCALAR chunk offset):
// a flat global_load needs 64-bit VALU address arithmetic per load, and VALU instructions take issue slots away
// from the MFMAs of the same SIMD -- in the isolated loop (tools/ubench/gemm_loop.hip) buffer loads issue at 65.7
// cycles per MFMA (the MFMA-only floor is 65.3), global loads at 68.7.
typedef float omds_f4 __attribute__((ext_vector_type(4)));
template <int MR, int NR>
__device__ __forceinline__ void load_chunk(const float* arow, __amdgpu_buffer_rsrc_t wrsrc, int wvoff, int c,
                                           float4 (&a)[MR], float4 (&w)[NR]) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const omds_f4 v = __builtin_bit_cast(omds_f4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, (j * (32 * 64) + c * 64) * 16, 0));
        w[j] = make_float4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) a[i] = *reinterpret_cast<const float4*>(arow + i * 32 * LDH + 8 * c);
}

// Software-pipelined by hand: the fragments of chunk c+1 (one global_load_dwordx4 per column block,
// one ds_read_b128 per row block) are issued BEFORE the 4*MR*NR MFMAs of chunk c and pinned there
// with sched_barrier (left alone, hipcc sinks the loads next to their use and waits vmcnt(0) per chunk).
template <int MR, int NR>
__device__ __forceinline__ void gemm256(const float* __restrict__ Hw, const float4* __restrict__ Wp, int cb0,
                                        int lane, f32x16 (&acc)[MR][NR]) {
    const float* arow = Hw + (lane & 31) * LDH + 4 * (lane >> 5);
    // cb0 derives from the wave index: wave-uniform, but only readfirstlane makes that provable to the compiler
    const float4* wbase = Wp + (size_t)__builtin_amdgcn_readfirstlane(cb0) * (32 * 64);
    const __amdgpu_buffer_rsrc_t wp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(wbase), 0, NR * 32 * 64 * 16, 0x00020000);
    const int wv = lane * 16;
    float4 a0[MR], a1[MR], w0[NR], w1[NR];
    load_chunk<MR, NR>(arow, wp, wv, 0, a0, w0);
    // One load (the weight fragment first, then the LDS reads) is slotted after every MFMA of the first half
    // of a cluster via sched_group_barrier: in the isolated loop (tools/ubench/gemm_loop.hip) this issues at
    // 68.6 cycles per MFMA against 70.3 for "all loads, then the cluster" and 75 for hipcc's own schedule.
#define OMDS_INTERLEAVE()                                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < NR; ++q_) {                               \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);                             \
    }                                                                                 \
    _Pragma("unroll") for (int q_ = 0; q_ < MR; ++q_) {                               \
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    }                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x8, 4 * MR * NR - NR - 2 * MR, 0);          \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 32; c += 2) {   // fully unrolled: every LDS / buffer offset is an immediate or an SGPR, no VALU in the loop
        load_chunk<MR, NR>(arow, wp, wv, c + 1, a1, w1);
        mfma_chunk<MR, NR>(a0, w0, acc);
        OMDS_INTERLEAVE()
        load_chunk<MR, NR>(arow, wp, wv, (c + 2) & 31, a0, w0);   // last iteration re-loads chunk 0 (harmless)
        mfma_chunk<MR, NR>(a1, w1, acc);
        OMDS_INTERLEAVE()
    }
#undef OMDS_INTERLEAVE
}

// C/D layout of the 32x32 MFMA: lane l, register r -> row (r&3) + 8(r>>2) + 4(l>>5), col l&31.
__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ------------------------------------------------------------------------------------------------
// [16 x 256] . [256 x 32] on v_mfma_f32_16x16x4_f32 -- the GEMM of the 16-row pass-2 tiles used for small batches,
// where a tile's chain of dependent GEMMs is the latency of the whole horizon step: half the rows = half the MFMA
// time per GEMM (3.4 us instead of 6.8 us on the four SIMDs of a CU).
//   Lane group g = l>>4 contracts, in steps 0..3 of chunk c, k = 16c + pa[g] + {0, 2, 8, 10} with pa = {0, 4, 1, 5}: the K
//   index of an MFMA runs over the lane groups, so the products of an output element are added in the order
//   16c + {0,4,1,5, 2,6,3,7, 8,12,9,13, 10,14,11,15} -- exactly the order of the 32-row kernels (gemm256: chunk of 8 k, step j
//   adds k = 8c'+j then 8c'+4+j).  Both MFMAs are plain fmaf chains in k order (tools/ubench/mfma_order.hip), so a row gets
//   the SAME bits from a 16-row and a 32-row tile.  A from LDS: two ds_read2_b32 per chunk; B packed alike (Wf16 / Wb16).
//   C/D: lane l, reg r -> row 4(l>>4)+r, col l&15.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pa16(int lane) { const int g = lane >> 4; return (g >> 1) + 4 * (g & 1); }
__device__ __forceinline__ float4 load_a16(const float* arow, int c) {   // arow = H + row * LDH + pa16(lane)
    return make_float4(arow[16 * c], arow[16 * c + 2], arow[16 * c + 8], arow[16 * c + 10]);
}
__device__ __forceinline__ void load_chunk16(const float* arow, __amdgpu_buffer_rsrc_t wrsrc, int wvoff, int c, float4& a, float4 (&w)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const omds_f4 v = __builtin_bit_cast(omds_f4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, (j * 16 + c) * 64 * 16, 0));
        w[j] = make_float4(v.x, v.y, v.z, v.w);
    }
    a = load_a16(arow, c);
}
__device__ __forceinline__ void mfma_chunk16(const float4& a, const float4 (&w)[2], f32x4 (&acc)[2]) {
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[0].x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[1].x, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[0].y, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[1].y, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[0].z, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[1].z, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[0].w, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[1].w, acc[1], 0, 0, 0);
}
// Wp = pack of one layer ([16 colblk16][16 kchunk][64 lane]); wave w produces columns 32w .. 32w+31 (blocks 2w, 2w+1)
__device__ __forceinline__ void gemm16(const float* __restrict__ Hs, const float4* __restrict__ Wp, int wave, int lane, f32x4 (&acc)[2]) {
    const float* arow = Hs + (lane & 15) * LDH + pa16(lane);
    const float4* wbase = Wp + (size_t)__builtin_amdgcn_readfirstlane(wave) * (2 * 16 * 64);
    const __amdgpu_buffer_rsrc_t wp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(wbase), 0, 2 * 16 * 64 * 16, 0x00020000);
    const int wv = lane * 16;
    float4 a0, a1, w0[2], w1[2];
    load_chunk16(arow, wp, wv, 0, a0, w0);
#define OMDS_INTERLEAVE16()                                 \
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);        \
    __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);       \
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);        \
    __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);       \
    __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);        \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);        \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
        load_chunk16(arow, wp, wv, c + 1, a1, w1);
        mfma_chunk16(a0, w0, acc);
        OMDS_INTERLEAVE16()
        load_chunk16(arow, wp, wv, (c + 2) & 15, a0, w0);
        mfma_chunk16(a1, w1, acc);
        OMDS_INTERLEAVE16()
    }
#undef OMDS_INTERLEAVE16
}


// ------------------------------------------------------------------------------------------------
// Layer 1: [rows x 32] . [32 x 256] over the tile's ENCODED INPUTS [x, sin x, cos x] (3d <= 30 features, zero-padded), which sit
// at positions 0..31 of the tile rows.  The reference computes the layer as one chain over the 3d features in their order
// [q, p, sin q, sin p, cos q, cos p] (network_macros_mod.py:139-141): joint and obstacle terms alternate along the chain, so the
// layer does NOT split into a rollout half plus an obstacle half without changing its bits -- it is a K = 32 product like
// every other layer (four k-chunks of the 32x32x2 shape, two of the 16x16x4 shape), 4 % of a launch's MFMA work.
// ------------------------------------------------------------------------------------------------
template <int MR, int NR>
__device__ __forceinline__ void gemm_k32(const float* __restrict__ Hw, const float4* __restrict__ W1f, int cb0, int lane, f32x16 (&acc)[MR][NR]) {
    const float* arow = Hw + (lane & 31) * LDH + 4 * (lane >> 5);
    const float4* wbase = W1f + (size_t)__builtin_amdgcn_readfirstlane(cb0) * (4 * 64);
    const __amdgpu_buffer_rsrc_t wp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(wbase), 0, NR * 4 * 64 * 16, 0x00020000);
    const int wv = lane * 16;
    float4 a[4][MR], w[4][NR];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const omds_f4 v = __builtin_bit_cast(omds_f4, __builtin_amdgcn_raw_buffer_load_b128(wp, wv, (j * (4 * 64) + c * 64) * 16, 0));
            w[c][j] = make_float4(v.x, v.y, v.z, v.w);
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) a[c][i] = *reinterpret_cast<const float4*>(arow + i * 32 * LDH + 8 * c);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) mfma_chunk<MR, NR>(a[c], w[c], acc);
}
// W1f16 = [16 colblk16][2 kchunk][64 lane]; wave w produces columns 32w .. 32w+31 (blocks 2w, 2w+1)
__device__ __forceinline__ void gemm16_k32(const float* __restrict__ Hs, const float4* __restrict__ W1f16, int wave, int lane, f32x4 (&acc)[2]) {
    const float* arow = Hs + (lane & 15) * LDH + pa16(lane);
    const float4* wbase = W1f16 + (size_t)__builtin_amdgcn_readfirstlane(wave) * (2 * 2 * 64);
    const __amdgpu_buffer_rsrc_t wp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(wbase), 0, 2 * 2 * 64 * 16, 0x00020000);
    float4 a[2], w[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const omds_f4 v = __builtin_bit_cast(omds_f4, __builtin_amdgcn_raw_buffer_load_b128(wp, lane * 16, (j * 2 + c) * 64 * 16, 0));
            w[c][j] = make_float4(v.x, v.y, v.z, v.w);
        }
        a[c] = load_a16(arow, c);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) mfma_chunk16(a[c], w[c], acc);
}

// ------------------------------------------------------------------------------------------------
// [4 NG x 256] . [256 x 64] on v_mfma_f32_4x4x1_16B_f32 with the A operand of ONE block broadcast to all 16 blocks
// (cbsz = 4, abid = g): D[i][lane] = fmaf(A[4g + i], B[lane], C[i][lane]) -- 4 rows x 64 columns x k = 1 per instruction at
// the full fp32 MFMA rate (tools/ubench/mfma_4x4.hip: 138-142 TFLOP/s by the wall clock from ONE wave per SIMD with 3-5
// independent chains, against 144-156 for 16x16x4), one fused multiply-add per element (76800 of 76800 random elements).  A
// GEMM on 4-row groups therefore has a row granularity of 4 where the 16x16x4 / 32x32x2 shapes have 16 / 32, and with K = 1 per
// instruction the k order of the 32-row kernels (8c + {0,4,1,5,2,6,3,7}) is just the issue order: bit-identical rows again.
//   A: lane l reads H[row l][k] (one register serves up to 16 row groups; abid picks the group);
//   B: lane l holds W[k][64 cw + l], packed so that one b128 is four consecutive k (MlpDev::Wb4: [layer][4 cb][64 kq][64 lane]).
//   acc[g][i] = element (row 4 g + i, column 64 cw + lane).
// One wave per SIMD does the work (a second one would fetch the same weight fragments a second time), so nothing but its own
// lookahead hides the L2 round trip: the weights stream through a ring of PD chunks (8 k each) that runs ACROSS the layers of
// the chain -- the slot of chunk c is asked for chunk c + PD of the same layer or chunk c + PD - 32 of the next one the moment
// it has been consumed.  sched_barrier pins every request where it is written: left to itself the scheduler sinks the loads
// to their uses (measured: 7.8 us per layer instead of 4.5).  "No next layer" is a VGPR offset of 0xffffffff: out of range
// for the buffer, the load returns zeros and moves no data.
// ------------------------------------------------------------------------------------------------
constexpr int W4_LAYER_BYTES = 4 * 64 * 64 * 16;
template <int PD>
struct W4Ring {
    float4 w[PD][2];
    __amdgpu_buffer_rsrc_t rs;
    int sbase;   // this wave's column block inside a layer pack, bytes
    int vlane;   // lane * 16
    __device__ __forceinline__ void bind(const float4* pack, int nl, int cw, int lane) {
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(pack), 0, nl * W4_LAYER_BYTES, 0x00020000);
        sbase = __builtin_amdgcn_readfirstlane(cw) * (64 * 64 * 16);
        vlane = lane * 16;
    }
    // chunk C (0..31) of layer l into slot S
    template <int S, int C>
    __device__ __forceinline__ void issue(int l) {
        const int voff = l >= 0 ? vlane + l * W4_LAYER_BYTES : -1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const omds_f4 v = __builtin_bit_cast(omds_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sbase + (2 * C + h) * 64 * 16, 0));
            w[S][h] = make_float4(v.x, v.y, v.z, v.w);
        }
    }
    // one half (four k) of chunk C of layer l into slot S
    template <int S, int C, int HALF>
    __device__ __forceinline__ void issue_half(int l) {
        const int voff = l >= 0 ? vlane + l * W4_LAYER_BYTES : -1;
        const omds_f4 v = __builtin_bit_cast(omds_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sbase + (2 * C + HALF) * 64 * 16, 0));
        w[S][HALF] = make_float4(v.x, v.y, v.z, v.w);
    }
    template <int C = 0>
    __device__ __forceinline__ void fill(int l) {
        if constexpr (C < PD) {
            issue<C, C>(l);
            __builtin_amdgcn_sched_barrier(0);
            fill<C + 1>(l);
        }
    }
};

template <int NG, int G>
__device__ __forceinline__ void g4_step(float av, float bv, f32x4 (&acc)[NG]) {
    if constexpr (G < NG) {
        acc[G] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[G], 4, G, 0);
        g4_step<NG, G + 1>(av, bv, acc);
    }
}
// Chunk C.  The requests that keep the pipeline full -- the two halves of the ring slot the PREVIOUS chunk has freed, the two
// LDS reads of the next chunk's A operand -- are each issued behind one group of MFMAs instead of in a cluster at the chunk
// boundary: an instruction issued in the shadow of a 2-pass MFMA costs nothing, a cluster of them between two MFMAs leaves
// the matrix pipe idle (9.7 cycles per MFMA with the cluster, against the 8.4 one wave can issue).
template <int NG, int PD, int C>
__device__ __forceinline__ void gemm4_chunks(const float* arow, W4Ring<PD>& R, int l, int l_next, float4 a_lo, float4 a_hi, f32x4 (&acc)[NG]) {
    if constexpr (C < 32) {   // chunk C = k 8C .. 8C+7, contracted in the order 8C + {0,4,1,5,2,6,3,7}
        constexpr int CP = C - 1 + PD;   // the chunk that slot (C - 1) % PD takes next
        float4 n_lo = a_lo, n_hi = a_hi;
        const float4 w_lo = R.w[C % PD][0], w_hi = R.w[C % PD][1];
        g4_step<NG, 0>(a_lo.x, w_lo.x, acc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (C >= 1) {
            if constexpr (CP < 32) R.template issue_half<(C - 1) % PD, CP, 0>(l);
            else R.template issue_half<(C - 1) % PD, CP - 32, 0>(l_next);
        }
        __builtin_amdgcn_sched_barrier(0);
        g4_step<NG, 0>(a_hi.x, w_hi.x, acc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (C >= 1) {
            if constexpr (CP < 32) R.template issue_half<(C - 1) % PD, CP, 1>(l);
            else R.template issue_half<(C - 1) % PD, CP - 32, 1>(l_next);
        }
        __builtin_amdgcn_sched_barrier(0);
        g4_step<NG, 0>(a_lo.y, w_lo.y, acc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (C + 1 < 32) n_lo = *reinterpret_cast<const float4*>(arow + 8 * (C + 1));
        __builtin_amdgcn_sched_barrier(0);
        g4_step<NG, 0>(a_hi.y, w_hi.y, acc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (C + 1 < 32) n_hi = *reinterpret_cast<const float4*>(arow + 8 * (C + 1) + 4);
        __builtin_amdgcn_sched_barrier(0);
        g4_step<NG, 0>(a_lo.z, w_lo.z, acc);
        g4_step<NG, 0>(a_hi.z, w_hi.z, acc);
        g4_step<NG, 0>(a_lo.w, w_lo.w, acc);
        g4_step<NG, 0>(a_hi.w, w_hi.w, acc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (C == 31) {   // the last chunk's own slot: the ring is whole again for the next layer
            constexpr int CL = 31 + PD - 32;
            R.template issue_half<31 % PD, CL, 0>(l_next);
            R.template issue_half<31 % PD, CL, 1>(l_next);
            __builtin_amdgcn_sched_barrier(0);
        }
        gemm4_chunks<NG, PD, C + 1>(arow, R, l, l_next, n_lo, n_hi, acc);
    }
}
// R holds chunks 0 .. PD-1 of layer l (filled, or left by the previous call); on return those of layer l_next
template <int NG, int PD>
__device__ __forceinline__ void gemm4(const float* __restrict__ Hs, W4Ring<PD>& R, int l, int l_next, int lane, f32x4 (&acc)[NG]) {
    static_assert(NG >= 1 && NG <= 8 && 32 % PD == 0, "row groups 0..7 (abid), ring slots that divide the 32 chunks");
    const float* arow = Hs + (lane & 31) * LDH;   // lanes >= 32 re-read rows 0..31: never selected (abid < 8)
    gemm4_chunks<NG, PD, 0>(arow, R, l, l_next, *reinterpret_cast<const float4*>(arow), *reinterpret_cast<const float4*>(arow + 4), acc);
}

template <int MT, int MR, int NR>
struct Geo {
    static constexpr int WM = MT / (32 * MR);
    static constexpr int WN = OMDS_NCB / NR;
    static constexpr int NW = WM * WN;
    static constexpr int NT = NW * 64;
    static_assert(WM >= 1 && WN >= 1 && WM * 32 * MR == MT && WN * NR == OMDS_NCB, "bad tile geometry");
};
// 16-row tile on v_mfma_f32_16x16x4 (gemm16): 8 waves, wave w owns columns 32w .. 32w+31 as two 16-column blocks
template <>
struct Geo<16, 1, 1> {
    static constexpr int WM = 1, WN = 8, NW = 8, NT = 512;
};

// ------------------------------------------------------------------------------------------------
// pass 1: all (rollout, obstacle) pairs -> min link distance
// ------------------------------------------------------------------------------------------------
// Diagnostic build (make timeline): workgroup phase timestamps, read by tools/pass1_timeline.py.  Compiles to nothing otherwise.
#ifdef OMDS_TIMELINE
#ifdef OMDS_TIMELINE_E   // per-WAVE stamps around one level's product and epilogue instead (tools/pass1_dyn_epilogue.py): [workgroup][wave][8]
#define OMDS_TL(i) do { } while (0)
#define OMDS_TLE(l, s) do { if (m.tl && (l) == OMDS_TIMELINE_E && (threadIdx.x & 63) == 0) m.tl[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (s)] = wall_clock64(); } while (0)
#else
#define OMDS_TL(i) do { if (m.tl && threadIdx.x == 0) m.tl[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#define OMDS_TLE(l, s) do { } while (0)
#endif
#define OMDS_TL_WAIT(what) asm volatile("s_waitcnt " what ::: "memory")
#else
#define OMDS_TL(i) do { } while (0)
#define OMDS_TLE(l, s) do { } while (0)
#define OMDS_TL_WAIT(what) do { } while (0)
#endif

// n / d for 0 <= n < 2^32 by multiplication (Granlund-Montgomery): q = (mulhi(mul, n) + n) >> shift with
// shift = ceil(log2 d), mul = floor(2^32 (2^shift - d) / d) + 1; n < 2^31 here, so the sum cannot overflow.
struct OmdsDivisor {
    unsigned mul;
    int shift;
    __device__ __forceinline__ unsigned div(unsigned n) const { return (__umulhi(mul, n) + n) >> shift; }
    static OmdsDivisor make(unsigned d) {
        OmdsDivisor r;
        r.shift = 0;
        while ((1ull << r.shift) < d) ++r.shift;
        r.mul = (unsigned)((((1ull << r.shift) - d) << 32) / d + 1);
        return r;
    }
};

// MODE 0: the tile covers rows row0 .. row0+MT-1 of the virtual rollout-major pair space (row = t*O + o) and writes
// Dmin[row].  MODE 1 (LIST; screening, screen_kernel.hip): the tile covers entries row0 .. of `rowlist`, each naming a pair
// t*O + o; the exact value, what pass 2's forward would compute for the row (pass-2 distance, arg-min link) and the ReLU
// masks go to ex->... per list entry, and max |screening value - exact value| to *maxerr_bits.  MODE 2 (fused small-O step,
// step_small.hip): rows as in mode 0, outputs as in mode 1 but indexed by the row within the tile (ex-> pointers are LDS
// arrays) and the masks stay in the tile's LDS block (maskS, behind rowIdx).  MODE 4 (audit sample, k_audit): rows as in
// mode 1, the rollout index of a pair running over all horizon steps' states; no outputs but max (ex->Da[entry] - exact value)
// into maxerr_bits[2].  MODE 3 (screened step of tanh networks): rows as in mode 1; the exact value overwrites the pair's
// screening value in Dmin, max |screening - exact| goes to *maxerr_bits, nothing else is kept.  MODE 5 (screened step of tanh
// networks with the derivative hand-over): mode 1 with the rows' activation derivatives 1 - h^2 of every hidden layer
// (ex->deriv[(level * cap + entry) * 256 + column]: 1 KB per entry and layer, written row-wise from the tile) in place of the
// ReLU masks -- what pass 2's forward would have left in its scratch, so that the tail runs the backward only (k_tail_sel).
// MODE 6 (the all-fp32 step without a second forward): rows and Dmin as in mode 0, and what mode 1 leaves per list entry --
// pass-2 distance, arg-min link, ReLU masks -- for EVERY pair, indexed by the pair (ex->dr / amin / mask [row]): the forward of
// the k rows a rollout ends up selecting has been computed here anyway, so the tail selects from Dmin and runs the backward only.
// The arithmetic of a row is the same in all forms and independent of the other rows of the tile: bit-identical results.
template <int MT, int MR, int NR, int ACT, int MODE = 0>
__device__ __forceinline__ void pass1_tile(const MlpDev& m, float* smem, const float* __restrict__ Fq,
                                           const float* __restrict__ Fp, const float* __restrict__ radius, int O,
                                           long long total_rows, uint32_t ignored, float* __restrict__ Dmin,
                                           const long long row0, const OmdsDivisor odiv,
                                           const int* __restrict__ rowlist = nullptr, unsigned* maxerr_bits = nullptr,
                                           const ExactOut* ex = nullptr) {
    constexpr bool LIST = MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5, EMIT = MODE == 1 || MODE == 2 || MODE == 6, DERIV = MODE == 5;
    constexpr bool EMITY = EMIT || DERIV;   // pass 2's distance and arg-min link per row (ex->dr, ex->amin)
    static_assert(!DERIV || MT == 16, "the derivative hand-over is written for the 16-row tiles of k_exact");
    using G = Geo<MT, MR, NR>;
    float* Hs = smem;                                           // [MT][LDH]
    float* rowRad = smem + MT * LDH;                            // [MT] obstacle radius of each row
    int* rowIdx = reinterpret_cast<int*>(rowRad + MT);          // [MT] LIST: pair index of each row
    uint32_t* maskS = reinterpret_cast<uint32_t*>(rowIdx + MT); // [MT][nhid][8] LIST: ReLU masks of the tile's rows
    const int nhid = m.nhh + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / G::WN, wn = wave % G::WN;
    // MODE 5: the activation derivatives of the tile's rows at `level`, row-wise (a wave per row, one float4 per lane: 1 KB
    // coalesced) from the activations the epilogue has just left in the tile; after the barrier behind the epilogue
    [[maybe_unused]] auto emit_deriv = [&](int level) {
        if constexpr (DERIV) {
            for (int r = wave; r < MT; r += G::NW) {
                const long long e_idx = row0 + r;
                if (e_idx < total_rows && e_idx < ex->cap) {
                    const float4 h = *reinterpret_cast<const float4*>(Hs + r * LDH + 4 * lane);
                    float4 dv;
                    dv.x = 1.f - h.x * h.x; dv.y = 1.f - h.y * h.y; dv.z = 1.f - h.z * h.z; dv.w = 1.f - h.w * h.w;
                    *reinterpret_cast<float4*>(ex->deriv + ((size_t)level * ex->cap + (size_t)e_idx) * OMDS_WIDTH + 4 * lane) = dv;
                }
            }
            if ((m.skip_mask >> level) & 1u) __syncthreads();   // the concatenated inputs are about to be written into these rows
        }
    };
    OMDS_TL(0);
#ifdef OMDS_TIMELINE
    if (m.tl && threadIdx.x == 0)   // HW_ID and XCC_ID: which CU this workgroup landed on
        m.tl[(size_t)blockIdx.x * 16 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |
                                            ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
#endif

    const int cb0 = wn * NR;
    constexpr int NBIAS = MT == 16 ? 2 : NR;   // distinct output columns per thread
    auto bias_col = [&](int j) { return MT == 16 ? wave * 32 + 16 * j + (lane & 15) : (cb0 + j) * 32 + (lane & 31); };
    float bcur[NBIAS];   // bias of the layer about to be multiplied, fetched one layer ahead (layer 1's: in flight across the gather)
#pragma unroll
    for (int j = 0; j < NBIAS; ++j) bcur[j] = m.b1[bias_col(j)];

    // ---- the tile's encoded inputs: row r = pair (t, o) gets Fq[t] | Fp[o] (each table is zero in the other's slots) at positions
    //      0..31.  A wave fills two rows per step, one per lane half; the row bookkeeping (rollout t, obstacle o, bounds) is
    //      wave-uniform and stays on the scalar unit, the fetches are buffer loads.  Kept short on purpose: next to a co-resident
    //      workgroup that streams MFMAs these instructions issue at roughly one per MFMA slot (tools/ubench/corun.hip), and the
    //      matrix pipe idles whenever both residents of a CU are outside their GEMM loops (tools/pass1_timeline.py). ----------
    {
        constexpr int IT = MT / G::NW;                          // rows per wave
        static_assert(IT % 2 == 0, "two rows per step");
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const unsigned row0u = (unsigned)row0;                  // the launcher keeps total_rows below 2^31
        const unsigned t0 = LIST ? 0u : odiv.div(row0u);        // row0 / O by multiplication (three scalar instructions)
        const int rows_here = (int)((total_rows - row0 < MT) ? (total_rows - row0) : MT);
        const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Fq) + (size_t)t0 * OMDS_FROW, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Fp), 0, 0x7fffffff, 0x00020000);
        const bool hi = lane >= 32;
        const int f4 = (lane & 31) * 4;
        int o = LIST ? 0 : (int)(row0u - t0 * (unsigned)O) + wv, dt = 0;   // un-listed rows: row r = wv + it * NW is pair (t0 + dt, o)
        if constexpr (!LIST) { while (o >= O) { o -= O; ++dt; } }
        float myrad = 0.f;                                      // lane it of the wave collects the radius of its row it
        int myidx = 0;
        uint32_t fv[IT / 2];
#pragma unroll
        for (int it = 0; it < IT; it += 2) {
            int offq[2], offp[2];
            bool ok[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                ok[h] = wv + (it + h) * G::NW < rows_here;
                float rad = 0.f;
                int idx = 0;
                offq[h] = 0; offp[h] = 0;
                if constexpr (LIST) {
                    if (ok[h]) {
                        idx = __builtin_amdgcn_readfirstlane(rowlist[row0 + wv + (it + h) * G::NW]);
                        const unsigned tt = odiv.div((unsigned)idx);
                        const int oo = (int)((unsigned)idx - tt * (unsigned)O);
                        offq[h] = (int)(tt * (OMDS_FROW * 4)); offp[h] = oo * (OMDS_FROW * 4);
                        rad = radius[oo];
                    }
                } else {
                    if (ok[h]) { offq[h] = dt * (OMDS_FROW * 4); offp[h] = o * (OMDS_FROW * 4); rad = radius[o]; }
                    o += G::NW;
                    while (o >= O) { o -= O; ++dt; }
                }
                const int rbits = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rad));   // wave-uniform: keep it in an SGPR for the asm
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(myrad) : "s"(rbits), "n"(it + h));
                if constexpr (LIST) {
                    const int ib = __builtin_amdgcn_readfirstlane(idx);
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(myidx) : "s"(ib), "n"(it + h));
                }
            }
            const uint32_t fq = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(qr, (hi ? offq[1] : offq[0]) + f4, 0, 0);
            const uint32_t fp = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(pr, (hi ? offp[1] : offp[0]) + f4, 0, 0);
            fv[it / 2] = (hi ? ok[1] : ok[0]) ? (fq | fp) : 0u;   // rows past the end: zero inputs (their results are never stored)
        }
        OMDS_TL_WAIT("vmcnt(0)");
        OMDS_TL(8);
        uint32_t* frow = reinterpret_cast<uint32_t*>(Hs) + (wv + (hi ? G::NW : 0)) * LDH + omds_kpos(lane & 31);
#pragma unroll
        for (int it = 0; it < IT; it += 2) frow[it * G::NW * LDH] = fv[it / 2];
        if (lane < IT) rowRad[wv + lane * G::NW] = myrad;
        if constexpr (LIST) { if (lane < IT) rowIdx[wv + lane * G::NW] = myidx; }
    }
    OMDS_TL_WAIT("lgkmcnt(0)");
    OMDS_TL(9);
    __syncthreads();
    OMDS_TL(1);
    // skip-connection networks: the encoded input of each row goes behind the activations of `level` (MlpDev::skip_mask)
    auto inject = [&](int level) {
        const int c0 = m.skip_col[level], F = 3 * m.d;
        for (int e = tid; e < MT * 32; e += G::NT) {
            const int r = e >> 5, f = e & 31;
            if (f < F) {
                uint32_t v = 0u;
                if (row0 + r < total_rows) {
                    unsigned pair;
                    if constexpr (LIST) pair = (unsigned)rowIdx[r];
                    else pair = (unsigned)row0 + (unsigned)r;
                    const unsigned t = odiv.div(pair), o = pair - t * (unsigned)O;
                    v = __builtin_bit_cast(uint32_t, Fq[(size_t)t * OMDS_FROW + f]) | __builtin_bit_cast(uint32_t, Fp[(size_t)o * OMDS_FROW + f]);
                }
                reinterpret_cast<uint32_t*>(Hs)[r * LDH + omds_kpos(c0 + f)] = v;
            }
        }
        __syncthreads();
    };

    // ---- layer 1 (l = -1, K = 32 over the encoded inputs) and the hidden -> hidden layers, each the reference's product: the
    //      accumulators start at ZERO, the k order is ascending (omds_kpos), the bias is added in the epilogue ----------------
    const float* Hw = Hs + (wm * MR * 32) * LDH;
    if constexpr (MT == 16) {
        const int scol0 = wave * 32 + omds_kpos(lane & 15);   // position of column wave*32 + 16 j + (lane & 15): + 16 j
        for (int l = -1; l < m.nhh; ++l) {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            float bnow[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bnow[j] = bcur[j];
            if (l + 1 < m.nhh) {
#pragma unroll
                for (int j = 0; j < 2; ++j) bcur[j] = m.bh[(l + 1) * OMDS_WIDTH + bias_col(j)];
            }
            if (l < 0) gemm16_k32(Hs, m.W1f16, wave, lane, acc);
            else gemm16(Hs, m.Wf16 + (size_t)l * (16 * 16 * 64), wave, lane, acc);
            __syncthreads();  // every wave has finished reading the tile
            if (l == 0) OMDS_TL(6);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][r] += bnow[j];
            if constexpr (EMIT) {   // ReLU masks of this level: row 4g + reg = ballot bits of lanes 16g .. 16g+15, columns 32 wave + 16 j + (lane & 15)
                uint32_t mw = 0;
                const int sh = 16 * ((lane & 15) >> 2);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const unsigned long long b0 = __ballot(acc[0][reg] > 0.f), b1 = __ballot(acc[1][reg] > 0.f);
                    if ((lane & 3) == reg) mw = (uint32_t)((b0 >> sh) & 0xffffu) | ((uint32_t)((b1 >> sh) & 0xffffu) << 16);
                }
                if (lane < 16) maskS[((size_t)lane * nhid + (l + 1)) * 8 + wave] = mw;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
                Hs[(4 * (lane >> 4) + (r & 3)) * LDH + scol0 + 16 * (r >> 2)] = actf(acc[r >> 2][r & 3], ACT);
            __syncthreads();
            emit_deriv(l + 1);
            if ((m.skip_mask >> (l + 1)) & 1u) inject(l + 1);
            OMDS_TL(2 + (l < 0 ? 0 : l));
        }
    } else
    for (int l = -1; l < m.nhh; ++l) {
        f32x16 acc[MR][NR];
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        float bnow[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) bnow[j] = bcur[j];
        if (l + 1 < m.nhh) {
#pragma unroll
            for (int j = 0; j < NR; ++j) bcur[j] = m.bh[(l + 1) * OMDS_WIDTH + (cb0 + j) * 32 + (lane & 31)];
        }
        if (l < 0) gemm_k32<MR, NR>(Hw, m.W1f, cb0, lane, acc);
        else gemm256<MR, NR>(Hw, m.Wf + (size_t)l * (OMDS_NCB * 32 * 64), cb0, lane, acc);
        __syncthreads();  // every wave has finished reading the tile
        if (l == 0) OMDS_TL(6);
        // one lane-dependent base address per column block; everything else of (row, col) is a compile-time offset, so the
        // 16 * MR stores of a block use immediate offsets (hipcc otherwise builds a VGPR address per row: VALU = matrix-pipe time)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            float* hb = Hs + (wm * MR * 32 + 4 * (lane >> 5)) * LDH + (cb0 + j) * 32 + omds_kpos(lane & 31);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float z = acc[i][j][r] + bnow[j];
                    hb[(i * 32 + (r & 3) + 8 * (r >> 2)) * LDH] = actf(z, ACT);
                    if constexpr (EMIT) {   // lanes 0-31 hold row (r&3) + 8(r>>2) of this 32-column block, lanes 32-63 that row + 4
                        const unsigned long long bal = __ballot(z > 0.f);
                        if (lane == 0) {
                            const int rr = wm * MR * 32 + i * 32 + (r & 3) + 8 * (r >> 2);
                            maskS[((size_t)rr * nhid + (l + 1)) * 8 + cb0 + j] = (uint32_t)bal;
                            maskS[((size_t)(rr + 4) * nhid + (l + 1)) * 8 + cb0 + j] = (uint32_t)(bal >> 32);
                        }
                    }
                }
        }
        __syncthreads();
        if ((m.skip_mask >> (l + 1)) & 1u) inject(l + 1);
        OMDS_TL(2 + (l < 0 ? 0 : l));
    }

    // ---- last layer (256 -> C, padded to 16) on v_mfma_f32_16x16x4_f32, 16 rows per wave.  Kept short on purpose
    //      (buffer loads with constant offsets, DPP min over the 16 link lanes, one 16-byte store per 4 rows): like
    //      the layer-1 build it mostly runs starved next to the other resident workgroup's GEMM ----------------
    const __amdgpu_buffer_rsrc_t wlr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(m.Wl), 0, 16 * 64 * 16, 0x00020000);
    for (int rb = __builtin_amdgcn_readfirstlane(wave); rb < MT / 16; rb += G::NW) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* arow = Hs + (rb * 16 + (lane & 15)) * LDH + pa16(lane);   // the k sequence of gemm16 (ascending k through omds_kpos)
        const int r4 = rb * 16 + 4 * (lane >> 4);
        [[maybe_unused]] float scr[4];   // LIST: the screening values of this lane's four rows (guard), in flight across the MFMA loop
        if constexpr (LIST) {
            const bool have_da = MODE == 4 || ex->Da != nullptr;   // the pair's screening value: the copy the selection took, or k_screen's matrix
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                scr[reg] = (row0 + r4 + reg < total_rows) ? (have_da ? ex->Da[row0 + r4 + reg] : Dmin[rowIdx[r4 + reg]]) : 0.f;
        }
        // 64 dependent MFMAs on one accumulator (the k order is the row's bits): the weights must not add a trip to L2 per
        // chunk on top.  Four chunks in flight, each slot re-armed when it has been consumed; sched_barrier keeps the requests
        // where they are written (left alone the loop was load, wait, four MFMAs, sixteen times).  k_exact 37.0 -> 36.5 us; 8 or 16
        // chunks in flight need its 80-register cap lifted (two workgroups per CU instead of three) and lose what they gain.
        constexpr int LPD = 4;
        omds_f4 wq[LPD];
#pragma unroll
        for (int c = 0; c < LPD; ++c) {
            wq[c] = __builtin_bit_cast(omds_f4, __builtin_amdgcn_raw_buffer_load_b128(wlr, lane * 16, c * 64 * 16, 0));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 a = load_a16(arow, c);
            const omds_f4 w = wq[c % LPD];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
            if (c + LPD < 16) wq[c % LPD] = __builtin_bit_cast(omds_f4, __builtin_amdgcn_raw_buffer_load_b128(wlr, lane * 16, (c + LPD) * 64 * 16, 0));
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef OMDS_TIMELINE
        asm volatile("s_nop 0" ::"v"(acc[0]) : "memory");
        OMDS_TL(10);
#endif
        // C/D layout 16x16: col = lane&15 (link), row = 4(lane>>4) + reg
        const int j = lane & 15;
        const float bj = m.bl[j];
        const bool pad = j >= m.C, ign = (ignored >> j) & 1u;
        const float4 rr = *reinterpret_cast<const float4*>(rowRad + r4);
        const float rad[4] = {rr.x, rr.y, rr.z, rr.w};
        float y[4];
        [[maybe_unused]] float ydr[4];
        [[maybe_unused]] int yam[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            if constexpr (EMITY) {   // pass 2's arg-min over ALL raw outputs (robot_sdf.py:155) and the distance of that link:
                // the minimum over the 16 link lanes by DPP, then the lowest lane holding it from a ballot (ties: lower link,
                // as a compare-and-swap reduction by (value, index) would give; eight dependent cross-lane shuffles shorter)
                const float yv = pad ? __builtin_inff() : acc[reg] + bj;
                float mn = yv;
                mn = fminf(mn, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mn), 0xB1, 0xF, 0xF, false)));
                mn = fminf(mn, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mn), 0x4E, 0xF, 0xF, false)));
                mn = fminf(mn, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mn), 0x141, 0xF, 0xF, false)));
                mn = fminf(mn, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mn), 0x140, 0xF, 0xF, false)));
                const unsigned long long eq = __ballot(yv == mn);
                const unsigned bits = (unsigned)(eq >> (lane & 48)) & 0xffffu;
                ydr[reg] = mn / m.out_div - rad[reg];
                yam[reg] = bits ? __builtin_ctz(bits) : 0;
            }
            float v = (acc[reg] + bj) / m.out_div - rad[reg];
            v = pad ? __builtin_inff() : (ign ? 1e6f : v);
            // min over the 16 lanes of the row group: xor 1, xor 2 inside the quad, then mirror within 8 and within 16
            v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
            v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
            v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)));
            v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)));
            y[reg] = v;
        }
        if constexpr (LIST) {
            if (j == 0) {
                float me = 0.f;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const long long e_idx = row0 + r4 + reg;
                    if (e_idx < total_rows) {
                        // candidates (MODE 1): |Da - D| -> maxerr_bits[0].  Audit sample (MODE 4: pairs that were NOT candidates):
                        // the one-sided Da - D the selection rule bounds by eps -> maxerr_bits[2]; a non-candidate whose
                        // screening value is too LOW is harmless, one too HIGH by more than eps could hide a top-k row
                        const float e = MODE == 4 ? scr[reg] - y[reg] : fabsf(scr[reg] - y[reg]);   // (modes 1, 3, 5: candidates)
                        if (!(e <= me)) me = (e == e) ? e : __builtin_inff();   // a NaN screening value (fp16 overflow) counts as an infinite error
                        if constexpr (MODE == 1 || MODE == 5) {
                            if (e_idx < ex->cap) { ex->D[e_idx] = y[reg]; ex->dr[e_idx] = ydr[reg]; ex->amin[e_idx] = yam[reg]; }
                        }
                        if constexpr (MODE == 3) Dmin[rowIdx[r4 + reg]] = y[reg];   // the exact value takes the screening value's place
                    }
                }
                if (me > 0.f) atomicMax(maxerr_bits + (MODE == 4 ? 2 : 0), __builtin_bit_cast(unsigned, me));   // non-negative floats order like their bits
            }
        } else if constexpr (MODE == 2) {
            if (j == 0) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) { ex->D[r4 + reg] = y[reg]; ex->dr[r4 + reg] = ydr[reg]; ex->amin[r4 + reg] = yam[reg]; }
            }
        } else
        if (j == 0) {
            const long long g = row0 + r4;
            if (g + 3 < total_rows) {
                *reinterpret_cast<float4*>(Dmin + g) = make_float4(y[0], y[1], y[2], y[3]);   // row0 and r4 are multiples of 4
                if constexpr (MODE == 6) {
                    *reinterpret_cast<float4*>(ex->dr + g) = make_float4(ydr[0], ydr[1], ydr[2], ydr[3]);
                    *reinterpret_cast<int4*>(ex->amin + g) = make_int4(yam[0], yam[1], yam[2], yam[3]);
                }
            } else {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (g + reg < total_rows) {
                        Dmin[g + reg] = y[reg];
                        if constexpr (MODE == 6) { ex->dr[g + reg] = ydr[reg]; ex->amin[g + reg] = yam[reg]; }
                    }
            }
        }
    }
    if constexpr (MODE == 1 || MODE == 6) {   // the tile's rows are consecutive entries (list entries / pairs): one contiguous block of masks
        __syncthreads();
        const long long words = (long long)((total_rows - row0 < MT) ? (total_rows - row0) : MT) * nhid * 8;
        for (int i = tid; i < words; i += G::NT) {
            const long long e_idx = row0 + i / (nhid * 8);
            if (e_idx < ex->cap) ex->mask[(size_t)row0 * nhid * 8 + i] = maskS[i];
        }
    }
#ifdef OMDS_TIMELINE
    __syncthreads();
    OMDS_TL(5);
#endif
}

// ------------------------------------------------------------------------------------------------
// pass 1 with PER-TILE COMPACTION of the hidden levels (k_pass1's 32- and 64-row tiles, ReLU networks without skips).
// A hidden unit whose activation is exactly zero in every row of the tile adds fmaf(0, w, acc) = acc to every chain of the next
// layer: leaving it out changes no bit, whatever the order -- as long as the units that ARE multiplied keep their ascending order.
// So every level is stored COMPACTED: a wave ballots which of its 32 units fired in the tile, the eight masks meet in LDS at the
// barrier every layer has anyway, and a unit's position is its rank among the tile's firing units (k-permuted by omds_kpos like
// every tile); dead columns are not stored at all.  The next product runs over ceil(T / 8) chunks, T = firing units of the tile
// (164 / 195 / 146 / 130 of 256 on the shelf task), and fetches its weight fragments BY UNIT: W^T rows (one per unit, 1 KB) named by
// the position -> unit table the epilogue leaves in LDS -- four coalesced 256-byte loads per chunk instead of one 1 KB fragment.
// Exact for every input: nothing is presumed about which units fire.
// ------------------------------------------------------------------------------------------------
// an LDS location as its 32-bit byte address, and a 16-byte read from one (ds_read_b128 with an immediate offset)
// (the host pass of the compiler parses device functions too: it has no LDS address space to cast into)
__device__ __forceinline__ uint32_t omds_lds_addr(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)p;
#else
    return 0u;
#endif
}
__device__ __forceinline__ float4 omds_lds_f4(uint32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
    const omds_f4 v = *(const __attribute__((address_space(3))) omds_f4*)a;
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return make_float4(0.f, 0.f, 0.f, 0.f);
#endif
}
constexpr int OMDS_IDS = OMDS_WIDTH + 64;   // entries of the position -> unit table (the pipelines read up to four 16-position chunks ahead)

template <int MR>
__device__ __forceinline__ void gemm_gather(const float* __restrict__ Hw, const uint32_t* __restrict__ idsS, const float* __restrict__ WT, int cb0,
                                            int lane, f32x16 (&acc)[MR][1], int nch, unsigned long long* tl_first = nullptr) {
    // LDS byte addresses, as integers: the two running addresses stay two registers with immediate offsets (derived from one another
    // by the optimiser they cost three additions per chunk)
    uint32_t arow = omds_lds_addr(Hw + (lane & 31) * LDH + 4 * (lane >> 5));
    uint32_t irow = omds_lds_addr(idsS + 4 * (lane >> 5));   // this lane half's four positions of a chunk: 8c + 4h .. + 3
    const __amdgpu_buffer_rsrc_t wt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(WT) + __builtin_amdgcn_readfirstlane(cb0) * 32, 0,
                                                                        OMDS_WIDTH * OMDS_WIDTH * 4, 0x00020000);
    const int colb = (lane & 31) * 4;
    float4 a0[MR], a1[MR];
    float b0[4], b1[4];
    uint4 i0, i1;
    auto loadA = [&](uint32_t ar, float4 (&a)[MR]) {
#pragma unroll
        for (int i = 0; i < MR; ++i) a[i] = omds_lds_f4(ar + i * 32 * LDH * 4);
    };
    auto loadI = [&](uint32_t ir) { return __builtin_bit_cast(uint4, omds_lds_f4(ir)); };
    auto loadB = [&](const uint4& id, float (&b)[4]) {   // W^T[unit][column]: byte offset unit * 1024 (the table's entry) + column * 4
        b[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wt, (int)id.x + colb, 0, 0));
        b[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wt, (int)id.y + colb, 0, 0));
        b[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wt, (int)id.z + colb, 0, 0));
        b[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wt, (int)id.w + colb, 0, 0));
    };
    auto mfma4 = [&](const float4 (&a)[MR], const float (&b)[4]) {
#pragma unroll
        for (int i = 0; i < MR; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[0], acc[i][0], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MR; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[1], acc[i][0], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MR; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[2], acc[i][0], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MR; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[3], acc[i][0], 0, 0, 0);
    };
    auto mfma4_first = [&](const float4 (&a)[MR], const float (&b)[4]) {   // the chain starts here: C = 0 is the instruction's literal, no accumulator is cleared
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MR; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[0], zero, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MR; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[1], acc[i][0], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MR; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[2], acc[i][0], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MR; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[3], acc[i][0], 0, 0, 0);
    };
    // three stages: the table entries of chunk c + 2 (LDS), the operands of chunk c + 1 (LDS, L2), the MFMAs of chunk c.
    // Chunk 0 is peeled so that no accumulator is cleared (a VALU instruction beside the MFMAs is not free: EXPERIMENTS.md R6.11, R6.13).
    const int n = __builtin_amdgcn_readfirstlane(nch);
    if (n == 0) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
        return;
    }
    i0 = loadI(irow);
    i1 = loadI(irow + 32);
    loadA(arow, a0);
    loadB(i0, b0);
    loadA(arow + 32, a1);
    loadB(i1, b1);
    i0 = loadI(irow + 64);
    __builtin_amdgcn_sched_barrier(0);
#ifdef OMDS_TIMELINE   // diagnostic build: when this product's first operands are here (tl_first: thread 0's slot)
    if (tl_first) { OMDS_TL_WAIT("vmcnt(4) lgkmcnt(1)"); *tl_first = wall_clock64(); }
#endif
    mfma4_first(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    // now (a1, b1) is chunk 1, i0 the table entries of chunk 2
    const int npair = (n - 1) >> 1;
#pragma unroll 1
    for (int p = 0; p < npair; ++p) {
        loadA(arow + 64, a0);
        loadB(i0, b0);
        i1 = loadI(irow + 96);
        __builtin_amdgcn_sched_barrier(0);
        mfma4(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        loadA(arow + 96, a1);
        loadB(i1, b1);
        i0 = loadI(irow + 128);
        __builtin_amdgcn_sched_barrier(0);
        mfma4(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        arow += 64;
        irow += 64;
        asm volatile("" : "+v"(arow), "+v"(irow));
    }
    if ((n - 1) & 1) mfma4(a1, b1);   // an even count: the last chunk is the one the final pair (or the prologue) fetched
}

#ifndef OMDS_DYN_PRIO_PROD
#define OMDS_DYN_PRIO_PROD 0
#endif
#ifndef OMDS_DYN_PRIO_EPI
#define OMDS_DYN_PRIO_EPI 3
#endif
#define OMDS_DYN_PRIO(p) __builtin_amdgcn_s_setprio(p)
template <int MT, int MR>
__device__ __forceinline__ void pass1_tile_dyn(const MlpDev& m, float* smem, const float* __restrict__ Fq, const float* __restrict__ Fp,
                                               const float* __restrict__ radius, int O, long long total_rows, uint32_t ignored,
                                               float* __restrict__ Dmin, const long long row0, const OmdsDivisor odiv) {
    using G = Geo<MT, MR, 1>;
    static_assert(G::NW == 8 && G::WM == 1, "eight waves, wave w owns the units 32 w .. 32 w + 31 of every level");
    float* Hs = smem;                                              // [MT][LDH]
    float* rowRad = smem + MT * LDH;                               // [MT]
    uint32_t* aliveS = reinterpret_cast<uint32_t*>(rowRad + MT);   // [8] which of a wave's 32 units fired in this tile at the current level
    uint32_t* idsS = aliveS + 8;                                   // [OMDS_IDS] position -> unit * 1024 of the current level
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int col = wv * 32 + (lane & 31);                         // the unit this thread's accumulators hold, at every level
    float bcur = m.b1[col];
    OMDS_DYN_PRIO(OMDS_DYN_PRIO_EPI);   // outside its products a wave's VALU / LDS work competes with the co-resident workgroup's MFMA stream for issue
    if (tid < OMDS_IDS - OMDS_WIDTH) idsS[OMDS_WIDTH + tid] = 0u;  // behind the table: valid offsets for the pipeline's look-ahead
    OMDS_TL(0);
#ifdef OMDS_TIMELINE
    if (m.tl && threadIdx.x == 0)   // HW_ID and XCC_ID: which CU this workgroup landed on
        m.tl[(size_t)blockIdx.x * 16 + 15] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
#endif

    // ---- the tile's encoded inputs (as pass1_tile, un-listed rows) -------------------------------------------------
    {
        constexpr int IT = MT / G::NW;
        const unsigned row0u = (unsigned)row0;
        const unsigned t0 = odiv.div(row0u);
        const int rows_here = (int)((total_rows - row0 < MT) ? (total_rows - row0) : MT);
        const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Fq) + (size_t)t0 * OMDS_FROW, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Fp), 0, 0x7fffffff, 0x00020000);
        const bool hi = lane >= 32;
        const int f4 = (lane & 31) * 4;
        int o = (int)(row0u - t0 * (unsigned)O) + wv, dt = 0;
        while (o >= O) { o -= O; ++dt; }
        // the radius of row wv + 8 j by lane j: one vector load (a scalar load per row is a chain of IT L2 round trips in front of the tile)
        float myrad = 0.f;
        if (lane < IT && wv + lane * G::NW < rows_here) {
            const unsigned g = row0u + (unsigned)(wv + lane * G::NW);
            myrad = radius[g - odiv.div(g) * (unsigned)O];
        }
        uint32_t fv[IT / 2];
#pragma unroll
        for (int it = 0; it < IT; it += 2) {
            int offq[2], offp[2];
            bool ok[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                ok[h] = wv + (it + h) * G::NW < rows_here;
                offq[h] = ok[h] ? dt * (OMDS_FROW * 4) : 0;
                offp[h] = ok[h] ? o * (OMDS_FROW * 4) : 0;
                o += G::NW;
                while (o >= O) { o -= O; ++dt; }
            }
            const uint32_t fq = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(qr, (hi ? offq[1] : offq[0]) + f4, 0, 0);
            const uint32_t fp = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(pr, (hi ? offp[1] : offp[0]) + f4, 0, 0);
            fv[it / 2] = (hi ? ok[1] : ok[0]) ? (fq | fp) : 0u;
        }
        uint32_t* frow = reinterpret_cast<uint32_t*>(Hs) + (wv + (hi ? G::NW : 0)) * LDH + omds_kpos(lane & 31);
#pragma unroll
        for (int it = 0; it < IT; it += 2) frow[it * G::NW * LDH] = fv[it / 2];
        if (lane < IT) rowRad[wv + lane * G::NW] = myrad;
    }
    __syncthreads();
    OMDS_TL(1);

    // ---- layer 1 (l = -1) and the hidden -> hidden layers on the compacted tile -----------------------------------------
    int T = 0;   // firing units of the level in the tile
    for (int l = -1; l < m.nhh; ++l) {
        f32x16 acc[MR][1];
        const float bnow = bcur;
        if (l + 1 < m.nhh) bcur = m.bh[(l + 1) * OMDS_WIDTH + col];
        OMDS_TLE(l, 0);
        OMDS_DYN_PRIO(OMDS_DYN_PRIO_PROD);
        if (l < 0) {
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
            gemm_k32<MR, 1>(Hs, m.W1f, wv, lane, acc);
        }
#ifdef OMDS_TIMELINE
        else gemm_gather<MR>(Hs, idsS, m.WhT + (size_t)l * (OMDS_WIDTH * OMDS_WIDTH), wv, lane, acc, (T + 7) >> 3,
                             (m.tl && threadIdx.x == 0 && l < 3) ? m.tl + (size_t)blockIdx.x * 16 + 11 + l : nullptr);
#else
        else gemm_gather<MR>(Hs, idsS, m.WhT + (size_t)l * (OMDS_WIDTH * OMDS_WIDTH), wv, lane, acc, (T + 7) >> 3);
#endif
        OMDS_DYN_PRIO(OMDS_DYN_PRIO_EPI);
        OMDS_TL(2 + 2 * (l + 1));   // this wave's share of the product done (diagnostic build)
        OMDS_TLE(l, 1);
        // bias (last, as the reference adds it), ReLU, and which of this wave's units fired in the tile (the column of lane l and of
        // lane l + 32 is the same unit).  v_max_f32 / v_max3_f32 written out: fmaxf() would canonicalise every operand first
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = acc[i][0][r] + bnow;
                asm("v_max_f32_e32 %0, 0, %1" : "=v"(acc[i][0][r]) : "v"(z));
            }
        float zm = 0.f;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(zm) : "v"(zm), "v"(acc[i][0][r]), "v"(acc[i][0][r + 1]));
        const unsigned long long bal = __ballot(zm > 0.f);
        const uint32_t mine = (uint32_t)bal | (uint32_t)(bal >> 32);
        if (lane == 0) aliveS[wv] = mine;
        OMDS_TLE(l, 2);
        __syncthreads();  // every wave has finished reading the tile and the table; the eight masks are there
        OMDS_TLE(l, 3);
        const uint4 mlo = *reinterpret_cast<const uint4*>(aliveS), mhi = *reinterpret_cast<const uint4*>(aliveS + 4);
        const uint32_t mk[8] = {mlo.x, mlo.y, mlo.z, mlo.w, mhi.x, mhi.y, mhi.z, mhi.w};
        int base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int cnt = __builtin_popcount(__builtin_amdgcn_readfirstlane(mk[w]));
            base += w < wv ? cnt : 0;
            total += cnt;
        }
        T = total;
        if (tid == 0) {   // statistics (omds_pass1_skip_stats): chunks the consumer of this level multiplies, firing units
            atomicAdd(m.skip_stats + 1 + (l + 1), (unsigned long long)(l + 1 < m.nhh ? (T + 7) >> 3 : (T + 15) >> 4));
            atomicAdd(m.skip_stats + 1 + (OMDS_MAX_HIDDEN + 1) + (l + 1), (unsigned long long)T);
            if (l < 0) atomicAdd(m.skip_stats, 1ull);
        }
        const bool alive = (mine >> (lane & 31)) & 1u;
        const int rank = base + __builtin_popcount(mine & ((1u << (lane & 31)) - 1u));
        if (alive) {
            const int pos = omds_kpos(rank);
            float* hb = Hs + (4 * (lane >> 5)) * LDH + pos;
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) hb[(i * 32 + (r & 3) + 8 * (r >> 2)) * LDH] = acc[i][0][r];
            if (lane < 32) idsS[pos] = (uint32_t)col * (OMDS_WIDTH * 4);
        }
        // zero columns up to the next multiple of 16 positions (the chunks of both consumers are whole); their table entries: unit 0
        const int Tp = (T + 15) & ~15;
        constexpr int CPP = G::NT / MT;   // columns one pass of the workgroup clears (8 or 16 of at most 15)
#pragma unroll
        for (int it = 0; it < (15 + CPP - 1) / CPP; ++it) {
            const int pz = T + tid / MT + it * CPP;
            if (pz < Tp) {
                Hs[(tid % MT) * LDH + omds_kpos(pz)] = 0.f;
                if (tid % MT == 0) idsS[omds_kpos(pz)] = 0u;
            }
        }
        OMDS_TLE(l, 4);
        __syncthreads();
        OMDS_TLE(l, 5);
        OMDS_TL(3 + 2 * (l + 1));
    }

    // ---- last layer (256 -> C, padded to 16) on v_mfma_f32_16x16x4_f32 over the compacted level, weights by unit from WlT ----
    const __amdgpu_buffer_rsrc_t wlt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(m.WlT), 0, OMDS_WIDTH * 16 * 4, 0x00020000);
    const int nch16 = (T + 15) >> 4;
    for (int rb = wv; rb < MT / 16; rb += G::NW) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* arow = Hs + (rb * 16 + (lane & 15)) * LDH + pa16(lane);
        const uint32_t* ir = idsS + pa16(lane);
        const int jb = (lane & 15) * 4;
        const int r4 = rb * 16 + 4 * (lane >> 4);
        constexpr int LPD = 4;
        float wq[LPD][4];
        auto loadW = [&](const uint32_t* ic, float (&w)[4]) {   // WlT[unit][link]: byte offset unit * 64 = table entry / 16
            w[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wlt, (int)(ic[0] >> 4) + jb, 0, 0));
            w[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wlt, (int)(ic[2] >> 4) + jb, 0, 0));
            w[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wlt, (int)(ic[8] >> 4) + jb, 0, 0));
            w[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wlt, (int)(ic[10] >> 4) + jb, 0, 0));
        };
        auto mfma4 = [&](const float* ac, const float (&w)[4]) {
            const float4 a = load_a16(ac, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[3], acc, 0, 0, 0);
        };
        // the weights run LPD chunks ahead of the chain, in a ROLLED loop of LPD chunks per trip (a guard per chunk in an unrolled
        // loop made every chunk wait for the newest fetch); the fetches behind the level read the table's look-ahead entries
#pragma unroll
        for (int c = 0; c < LPD; ++c) loadW(ir + 16 * c, wq[c]);
        __builtin_amdgcn_sched_barrier(0);
        int c = 0;
#pragma unroll 1
        for (; c + LPD <= nch16; c += LPD) {
#pragma unroll
            for (int q = 0; q < LPD; ++q) {
                mfma4(arow + 16 * q, wq[q]);
                loadW(ir + 16 * (LPD + q), wq[q]);
                __builtin_amdgcn_sched_barrier(0);
            }
            arow += 16 * LPD;
            ir += 16 * LPD;
        }
#pragma unroll
        for (int q = 0; q < LPD - 1; ++q)
            if (c + q < nch16) mfma4(arow + 16 * q, wq[q]);
        const int j = lane & 15;
        const float bj = m.bl[j];
        const bool pad = j >= m.C, ign = (ignored >> j) & 1u;
        const float4 rr = *reinterpret_cast<const float4*>(rowRad + r4);
        const float rad[4] = {rr.x, rr.y, rr.z, rr.w};
        float y[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float v = (acc[reg] + bj) / m.out_div - rad[reg];
            v = pad ? __builtin_inff() : (ign ? 1e6f : v);
            v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
            v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
            v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)));
            v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)));
            y[reg] = v;
        }
        if (j == 0) {
            const long long g = row0 + r4;
            if (g + 3 < total_rows) {
                *reinterpret_cast<float4*>(Dmin + g) = make_float4(y[0], y[1], y[2], y[3]);
            } else {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (g + reg < total_rows) Dmin[g + reg] = y[reg];
            }
        }
    }
#ifdef OMDS_TIMELINE
    __syncthreads();
    OMDS_TL(10);
#endif
}

// k smallest entries of row[0..O) in ascending order (ties by lower index), one wave per row.
// emit(j, index) is called by lane 0.  Rows of up to 512 entries are held in registers.
// In-register rows: every entry becomes ONE 64-bit key, (order-preserving image of the value) << 32 | index, so that "smaller value,
// then lower index" is an unsigned comparison, "after the previous pick" is key > previous key, and a round is a local minimum over
// the lane's keys + a wave minimum on DPP row operations (quad xor 1 / 2, half-row and row mirror, row broadcasts 15 / 31: no LDS
// crossbar in the chain) -- a third of the cycles of the (value, index) butterfly of ds_bpermutes it replaces.  -0 counts as +0
// (they compare equal, the index decides); a NaN is never picked (its key is the "nothing" key); rows with fewer than k pickable
// entries repeat index 0, as before.
__device__ __forceinline__ unsigned long long omds_dpp_min_u64(unsigned long long v, unsigned long long o) { return o < v ? o : v; }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long omds_dpp_step_min_u64(unsigned long long v) {
    const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return omds_dpp_min_u64(v, ((unsigned long long)ohi << 32) | olo);
}
// the minimum over the wave, in every lane's copy of lane 63 (returned as a uniform value)
__device__ __forceinline__ unsigned long long omds_wave_min_u64(unsigned long long v) {
    v = omds_dpp_step_min_u64<0xB1, 0xF>(v);    // quad_perm [1, 0, 3, 2]
    v = omds_dpp_step_min_u64<0x4E, 0xF>(v);    // quad_perm [2, 3, 0, 1]
    v = omds_dpp_step_min_u64<0x141, 0xF>(v);   // row_half_mirror
    v = omds_dpp_step_min_u64<0x140, 0xF>(v);   // row_mirror: every lane of a 16-lane row holds the row's minimum
    v = omds_dpp_step_min_u64<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v = omds_dpp_step_min_u64<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's minimum
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}
template <typename Emit>
__device__ __forceinline__ void topk_row(const float* __restrict__ row, int O, int k, int lane, Emit emit) {
    constexpr int NV = 8;
    constexpr unsigned long long NONE = ~0ull;
    if (O <= 64 * NV) {
        unsigned long long key[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int o = lane + 64 * i;
            key[i] = NONE;
            if (o < O) {
                const float x = row[o] + 0.f;   // -0 -> +0
                const unsigned u = __builtin_bit_cast(unsigned, x);
                const unsigned ord = u ^ ((unsigned)((int)u >> 31) | 0x80000000u);   // x < y  <=>  ord(x) < ord(y)
                if (x == x) key[i] = ((unsigned long long)ord << 32) | (unsigned)o;
            }
        }
        unsigned long long prev = 0ull;
        bool first = true;
        for (int j = 0; j < k; ++j) {
            unsigned long long best = NONE;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const bool after = first || key[i] > prev;
                best = (after && key[i] < best) ? key[i] : best;
            }
            best = omds_wave_min_u64(best);
            if (lane == 0) emit(j, best == NONE ? 0 : (int)(unsigned)best);
            prev = best;
            first = false;
        }
        return;
    }
    float pv = -__builtin_inff();
    int pi = -1;
    for (int j = 0; j < k; ++j) {
        float bv = __builtin_inff();
        int bi = 0x7fffffff;
        for (int o = lane; o < O; o += 64) {
            const float x = row[o];
            const bool after = (x > pv) || (x == pv && o > pi);
            if (after && ((x < bv) || (x == bv && o < bi))) { bv = x; bi = o; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if ((ov < bv) || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (bi == 0x7fffffff) bi = 0;
        if (lane == 0) emit(j, bi);
        pv = bv;
        pi = bi;
    }
}

constexpr int P2_MT = 32;
constexpr int P2_NT = 512;


struct P2Smem {
    float* Hs;        // [32][LDH]
    float* gf;        // [32][33] feature gradients
    uint16_t* maskL;  // [nhh+1][512] ReLU masks, 16 bits per thread and layer
    uint32_t* maskG4; // the same storage seen by the 4-row-group backward: [nhh+1][256 columns], bit e = row e
    int* rowT;        // [32] rollout of each row (-1: padding row)
    int* rowO;        // [32] obstacle of each row
    int* rowMin;      // [32] arg-min link of each row
    const int* rowE = nullptr;   // tanh derivative hand-over (k_tail_sel): [32] list entry whose derivative rows each backward row reads
                                 // (-1: padding row), dscr then being ExactOut::deriv with dlayer = cap * 256; nullptr: row S0 + r of dscr
};

// Geometry of the per-thread accumulator values of a pass-2 tile: ROWS = 32 uses the 32x32x2 MFMA (16 values per thread,
// one column), ROWS = 16 the 16x16x4 MFMA (two 16-column blocks -> 8 values per thread, two columns).
template <int ROWS>
struct P2Geo {
    static constexpr int NV = ROWS == 32 ? 16 : 8;
    static constexpr int NB = ROWS == 32 ? 1 : 2;    // distinct columns per thread
    static __device__ __forceinline__ int row(int r, int lane) { return ROWS == 32 ? crow(r, lane) : 4 * (lane >> 4) + (r & 3); }
    static __device__ __forceinline__ int col(int r, int wave, int lane) {
        return ROWS == 32 ? wave * 32 + (lane & 31) : wave * 32 + 16 * (r >> 2) + (lane & 15);
    }
    static __device__ __forceinline__ int blk(int r) { return ROWS == 32 ? 0 : (r >> 2); }
    // position of that column in the k-permuted tile (omds_kpos keeps groups of eight columns together)
    static __device__ __forceinline__ int pos(int r, int wave, int lane) {
        return ROWS == 32 ? wave * 32 + omds_kpos(lane & 31) : wave * 32 + 16 * (r >> 2) + omds_kpos(lane & 15);
    }
};

// One [ROWS x 256] . [256 x 256] GEMM of pass 2 (forward pack or transposed pack of layer l; l = -1: layer 1 over the encoded
// inputs at positions 0..31, K = 32); out[r] in P2Geo order.  The accumulators start at zero in every product, forward and
// backward (the reference's chains; the forward's bias is added by the caller) -- the same arithmetic as pass1_tile, so the
// forward of a row is bit-identical in pass 1 and pass 2.
template <int ROWS>
__device__ __forceinline__ void p2_gemm(const float* Hs, const MlpDev& m, int l, bool backward, int wave, int lane,
                                        float (&out)[P2Geo<ROWS>::NV]) {
    if constexpr (ROWS == 32) {
        f32x16 acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
        if (l < 0) gemm_k32<1, 1>(Hs, m.W1f, wave, lane, acc);
        else gemm256<1, 1>(Hs, (backward ? m.Wb : m.Wf) + (size_t)l * (OMDS_NCB * 32 * 64), wave, lane, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) out[r] = acc[0][0][r];
    } else {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (l < 0) gemm16_k32(Hs, m.W1f16, wave, lane, acc);
        else gemm16(Hs, (backward ? m.Wb16 : m.Wf16) + (size_t)l * (16 * 16 * 64), wave, lane, acc);
#pragma unroll
        for (int r = 0; r < 8; ++r) out[r] = acc[r >> 2][r & 3];
    }
}

template <int ACT, int ROWS>
__device__ __forceinline__ void pass2_backward(const MlpDev& m, const P2Smem& sm, const float* __restrict__ xyzr, int R0,
                                               int total_rows, const float* __restrict__ qT, int ldq, float* gradx, int dbase,
                                               float* __restrict__ dscr, size_t dlayer, int S0, int dbg = 0);
template <int ROWS>
__device__ __forceinline__ void p2_backward_first(const MlpDev& m, const P2Smem& sm, const float* __restrict__ xyzr, int R0,
                                                  int total_rows, const float* __restrict__ qT, int ldq, float* gradx, int dbase, int dbg = 0);

// The last layer of pass 2 on the tile in LDS (v_mfma_f32_16x16x4, waves 0 .. ROWS / 16 - 1), the arg-min over ALL raw outputs
// (robot_sdf.py:155) and the distance of that link: sm.rowMin[r], drow[dbase + r]; optionally the raw outputs, the arg-min per global
// row, and the pass-1 value of each row (d1row: min over the un-ignored links of y / out_div - radius, MPPI.py:236-242).  seed_col >= 0:
// that output column instead of the arg-min one.  Shared by pass2_body and pass2_body_g4 (32 rows there, 20 of them real).
template <int ROWS>
__device__ __forceinline__ void p2_last_layer(const MlpDev& m, const float* Hs, int* rowMin, const int* rowO, const float* __restrict__ radius,
                                              int R0, int total_rows, float* drow, int dbase, float* __restrict__ yraw,
                                              int32_t* __restrict__ minidx, float* d1row, uint32_t ignored, int seed_col) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave < ROWS / 16) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* arow = Hs + (wave * 16 + (lane & 15)) * LDH + pa16(lane);
        const int j = lane & 15;
        const float bj = m.bl[j];
        // all 16 weight fragments in flight at once: the MFMA chain below is latency-bound otherwise
        {
            float4 wl[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) wl[c] = m.Wl[c * 64 + lane];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float4 a = load_a16(arow, c);
                const float4 w = wl[c];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = wave * 16 + 4 * (lane >> 4) + reg;
            const int R = R0 + r;
            const float y = acc[reg] + bj;
            if (yraw != nullptr && R < total_rows) yraw[(size_t)R * OMDS_CPAD + j] = (j < m.C) ? y : 0.f;
            float bv = (j < m.C && (seed_col < 0 || j == seed_col)) ? y : __builtin_inff();
            int bi = j;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const float ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                if ((ov < bv) || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (d1row != nullptr) {
                float v = y / m.out_div - radius[rowO[r]];
                v = (j >= m.C) ? __builtin_inff() : (((ignored >> j) & 1u) ? 1e6f : v);
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) v = fminf(v, __shfl_xor(v, off));
                if (j == 0) d1row[r] = v;
            }
            if (j == 0) {
                rowMin[r] = bi;
                if (R < total_rows) {
                    drow[dbase + r] = bv / m.out_div - radius[rowO[r]];
                    if (minidx != nullptr) minidx[R] = bi;
                }
            }
        }
    }
}

// Body of pass 2 for the ROWS rows described by sm.rowT / sm.rowO (already in LDS, barrier done by the
// caller).  R0 = global index of row 0 (tanh scratch, yraw, minidx); outputs go to
// gradx[(dbase + row) * d + j] and drow[dbase + row] (global memory or LDS).
template <int ACT, int ROWS = 32>
__device__ __forceinline__ void pass2_body(const MlpDev& m, const P2Smem& sm, const float* __restrict__ Fq,
                                           const float* __restrict__ Fp, const float* __restrict__ radius,
                                           const float* __restrict__ xyzr, int R0, int total_rows,
                                           const float* __restrict__ qT, int ldq, float* gradx, float* drow, int dbase,
                                           float* __restrict__ yraw, int32_t* __restrict__ minidx,
                                           float* __restrict__ dscr, size_t dlayer, int S0, int dbg = 0,
                                           float* d1row = nullptr, uint32_t ignored = 0, int seed_col = -1) {
    // seed_col >= 0: the backward starts from that output column instead of the arg-min one (one Jacobian column,
    // robot_sdf.py:92-100); minidx / drow then describe that column
    // d1row (optional, [ROWS]): the pass-1 value of each row, min over the un-ignored links of y / out_div - radius
    // (MPPI.py:236-242), computed from the same last-layer outputs -- bit-identical to pass1_tile's Dmin for 32-row tiles
    // dbg: timing experiments only (return after a stage).  S0 = first row of this workgroup's private slot in the tanh scratch
    using G = P2Geo<ROWS>;
    constexpr int NV = G::NV;
    float* Hs = sm.Hs;
    uint16_t* maskL = sm.maskL;
    int* rowT = sm.rowT;
    int* rowO = sm.rowO;
    int* rowMin = sm.rowMin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr bool relu = ACT == OMDS_ACT_RELU;

    float bnext[G::NB];   // bias of the layer about to be multiplied, fetched one layer ahead (no L2 round trip in front of its epilogue)
#pragma unroll
    for (int jb = 0; jb < G::NB; ++jb) bnext[jb] = m.b1[G::col(4 * jb, wave, lane)];
    // ---- the rows' encoded inputs [x, sin x, cos x] at positions 0..31: Fq[t] | Fp[o] (padding rows: zero) ----------------
    for (int e = tid; e < ROWS * 32; e += P2_NT) {
        const int r = e >> 5, f = e & 31, t = rowT[r];
        uint32_t v = 0u;
        if (t >= 0) v = __builtin_bit_cast(uint32_t, Fq[(size_t)t * OMDS_FROW + f]) | __builtin_bit_cast(uint32_t, Fp[(size_t)rowO[r] * OMDS_FROW + f]);
        reinterpret_cast<uint32_t*>(Hs)[r * LDH + omds_kpos(f)] = v;
    }
    __syncthreads();
    // skip-connection networks: the encoded input of each row behind the activations of `level` (MlpDev::skip_mask)
    auto inject = [&](int level) {
        const int c0 = m.skip_col[level], F = 3 * m.d;
        for (int e = tid; e < ROWS * 32; e += P2_NT) {
            const int r = e >> 5, f = e & 31;
            if (f < F) {
                const int t = rowT[r];
                uint32_t v = 0u;
                if (t >= 0) v = __builtin_bit_cast(uint32_t, Fq[(size_t)t * OMDS_FROW + f]) | __builtin_bit_cast(uint32_t, Fp[(size_t)rowO[r] * OMDS_FROW + f]);
                reinterpret_cast<uint32_t*>(Hs)[r * LDH + omds_kpos(c0 + f)] = v;
            }
        }
        __syncthreads();
    };

    // ---- forward: layer 1 (l = -1) and the hidden -> hidden layers, each the reference's product (from zero, ascending k, bias
    //      last); level l + 1 of the masks / derivatives ---------------------------------------------------------------------
    for (int l = -1; l < m.nhh; ++l) {
        float bv[G::NB];
#pragma unroll
        for (int jb = 0; jb < G::NB; ++jb) bv[jb] = bnext[jb];
        if (l + 1 < m.nhh) {
#pragma unroll
            for (int jb = 0; jb < G::NB; ++jb) bnext[jb] = m.bh[(l + 1) * OMDS_WIDTH + G::col(4 * jb, wave, lane)];
        }
        float acc[NV];
        p2_gemm<ROWS>(Hs, m, l, false, wave, lane, acc);
        __syncthreads();
        uint32_t bits = 0;
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int row = G::row(r, lane), col = G::col(r, wave, lane);
            const float z = acc[r] + bv[G::blk(r)];
            bits |= (z > 0.f ? 1u : 0u) << r;
            const float h = actf(z, ACT);
            Hs[row * LDH + G::pos(r, wave, lane)] = h;
            if (!relu) dscr[(l + 1) * dlayer + (size_t)(S0 + row) * OMDS_WIDTH + col] = 1.f - h * h;
        }
        maskL[(l + 1) * P2_NT + tid] = (uint16_t)bits;
        __syncthreads();
        if (dbg == 10 && l < 0) return;
        if ((m.skip_mask >> (l + 1)) & 1u) inject(l + 1);
    }

    if (dbg == 11) return;
    // ---- last layer, arg-min over ALL raw outputs (robot_sdf.py:155), distance of that link -------
    p2_last_layer<ROWS>(m, Hs, rowMin, rowO, radius, R0, total_rows, drow, dbase, yraw, minidx, d1row, ignored, seed_col);
    __syncthreads();
    if (dbg == 12) return;
    pass2_backward<ACT, ROWS>(m, sm, xyzr, R0, total_rows, qT, ldq, gradx, dbase, dscr, dlayer, S0, dbg);
}

// 1 - h^2 of (row, col) at one hidden level: from the workgroup's own scratch rows (pass 2 behind its own forward) or, with
// sm.rowE, from the derivative rows k_exact left for the list entry of each backward row (a padding row multiplies by zero)
__device__ __forceinline__ float p2_tanh_deriv(const P2Smem& sm, const float* __restrict__ dscr, size_t level_off, int S0, int row, int col) {
    if (sm.rowE == nullptr) return dscr[level_off + (size_t)(S0 + row) * OMDS_WIDTH + col];
    const int e = sm.rowE[row];   // k_exact wrote its rows as they sit in the tile: column col at position omds_kpos(col)
    return e >= 0 ? dscr[level_off + (size_t)e * OMDS_WIDTH + omds_kpos(col)] : 0.f;
}

// The backward half of pass 2: from sm.rowMin (arg-min link of each row), the activation derivatives -- sm.maskL (ReLU:
// 16 bits per thread and layer in the MFMA C layout) or dscr (tanh) -- and sm.rowT / sm.rowO to the input gradients
// gradx[(dbase + row) * d + j].  pass2_body runs it behind its own forward; the screened step (k_tail_sel) runs it on masks
// that k_exact produced when it evaluated the candidates.
template <int ACT, int ROWS>
__device__ __forceinline__ void pass2_backward(const MlpDev& m, const P2Smem& sm, const float* __restrict__ xyzr, int R0,
                                               int total_rows, const float* __restrict__ qT, int ldq, float* gradx, int dbase,
                                               float* __restrict__ dscr, size_t dlayer, int S0, int dbg) {
    using G = P2Geo<ROWS>;
    constexpr int NV = G::NV;
    float* Hs = sm.Hs;
    float* gf = sm.gf;
    uint16_t* maskL = sm.maskL;
    int* rowMin = sm.rowMin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr bool relu = ACT == OMDS_ACT_RELU;

    // skip-connection networks: the gradient that arrives at the concatenated input columns of a level is a direct
    // contribution to the feature gradient; it is collected in gf (one owner thread per element and level)
    const int F3 = 3 * m.d;
    if (m.skip_mask) {
        for (int e = tid; e < ROWS * 33; e += P2_NT) gf[e] = 0.f;
        __syncthreads();
    }
    // ---- backward seed: dy[minIdx]/dH_last = Wlast[minIdx], masked by the last hidden layer ---------
    {
        const uint32_t bits = maskL[m.nhh * P2_NT + tid];
        const bool cap = (m.skip_mask >> m.nhh) & 1u;
        const int c0 = m.skip_col[m.nhh];
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int row = G::row(r, lane), col = G::col(r, wave, lane);
            const float g = m.Wlraw[(size_t)rowMin[row] * OMDS_WIDTH + col];
            if (cap && col >= c0 && col < c0 + F3) gf[row * 33 + col - c0] += g;
            const float dv = relu ? (((bits >> r) & 1u) ? 1.f : 0.f) : p2_tanh_deriv(sm, dscr, m.nhh * dlayer, S0, row, col);
            Hs[row * LDH + G::pos(r, wave, lane)] = g * dv;
        }
    }
    __syncthreads();
    OMDS_TL_STAMP(4);
    if (dbg == 13) return;

    // ---- backward through the hidden -> hidden layers ------------------------------------------
    for (int l = m.nhh - 1; l >= 0; --l) {
        float acc[NV];
        p2_gemm<ROWS>(Hs, m, l, true, wave, lane, acc);
        if (l == m.nhh - 1) OMDS_TL_STAMP(5);
        __syncthreads();
        if (l == m.nhh - 1) OMDS_TL_STAMP(6);
        const uint32_t bits = maskL[l * P2_NT + tid];
        const bool cap = (m.skip_mask >> l) & 1u;
        const int c0 = m.skip_col[l];
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int row = G::row(r, lane), col = G::col(r, wave, lane);
            const float dv = relu ? (((bits >> r) & 1u) ? 1.f : 0.f) : p2_tanh_deriv(sm, dscr, l * dlayer, S0, row, col);
            if (cap && col >= c0 && col < c0 + F3) gf[row * 33 + col - c0] += acc[r];
            Hs[row * LDH + G::pos(r, wave, lane)] = acc[r] * dv;
        }
        __syncthreads();
        if (l == m.nhh - 1) OMDS_TL_STAMP(7);
    }
    OMDS_TL_STAMP(8);

    if (dbg == 14) return;
    p2_backward_first<ROWS>(m, sm, xyzr, R0, total_rows, qT, ldq, gradx, dbase, dbg);
}

// d y / d x of one input from the gradients at its three encoded features, as torch's autograd accumulates them for
// x_nerf = cat(x, sin x, cos x) (network_macros_mod.py:139-140): (g_x + g_cos * (-sin x)) + g_sin * cos x, every operation rounded
// on its own (established bit for bit against the reference's gradients, tools/studies/assoc_order_study.py --vjp)
__device__ __forceinline__ float pe_chain_rule(float gx, float gsin, float gcos, float x) {
    return __fadd_rn(__fadd_rn(gx, __fmul_rn(gcos, -omds_sinf(x))), __fmul_rn(gsin, omds_cosf(x)));
}

// The first layer's backward as the reference computes it (robot_sdf.py:153-158: (g * mask) @ W1): for every (row, feature) ONE fmaf
// chain over the 256 hidden units in ascending order from zero.  [ROWS x 256] . [256 x 32] on v_mfma_f32_16x16x4: wave w owns the
// 16 x 16 block (rows 16 (w >> 1), features 16 (w & 1)) and runs the whole k range -- 64 dependent MFMAs, 1 us; splitting k over
// the waves (the round-1 form) was 0.5 us shorter and summed eight partial chains.  Hs = the gradient at the first layer's
// pre-activations (k-permuted tile, read like gemm16); out[row * 33 + f] = the chain (+ prior[row * 33 + f], the skip
// concatenations' direct contributions, when prior != nullptr; out may alias prior).
template <int ROWS>
__device__ __forceinline__ void first_layer_backward(const MlpDev& m, const float* Hs, float* out, const float* prior) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave < ROWS / 8) {
        const int rb = wave >> 1, jb = wave & 1;
        const float* arow = Hs + (rb * 16 + (lane & 15)) * LDH + pa16(lane);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float4 wq[16];   // all 16 weight fragments in flight at once: the chain below is latency-bound otherwise
#pragma unroll
        for (int c = 0; c < 16; ++c) wq[c] = m.W1b16[(c * 2 + jb) * 64 + lane];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 a = load_a16(arow, c);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wq[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wq[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wq[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wq[c].w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // C/D layout 16x16: column (feature) lane & 15, row 4 (lane >> 4) + r
            const int e = (rb * 16 + 4 * (lane >> 4) + r) * 33 + 16 * jb + (lane & 15);
            out[e] = prior ? acc[r] + prior[e] : acc[r];
        }
    }
}

// The tail of the pass-2 backward: from the gradient at the first layer's pre-activations (sm.Hs) to the input gradients.
template <int ROWS>
__device__ __forceinline__ void p2_backward_first(const MlpDev& m, const P2Smem& sm, const float* __restrict__ xyzr, int R0,
                                                  int total_rows, const float* __restrict__ qT, int ldq, float* gradx, int dbase, int dbg) {
    float* gf = sm.gf;
    int* rowT = sm.rowT;
    int* rowO = sm.rowO;
    const int tid = threadIdx.x;
    first_layer_backward<ROWS>(m, sm.Hs, gf, m.skip_mask ? gf : nullptr);
    __syncthreads();
    if (dbg == 15) return;
    // ---- positional-encoding chain rule -------------------------------------------------------------------------------
    const int d = m.d, n = m.n_dof;
    if (tid < ROWS * d) {
        const int row = tid / d, jj = tid - row * d;
        const int R = R0 + row;
        if (R < total_rows && rowT[row] >= 0) {
            const float x = (jj < n) ? qT[(size_t)jj * ldq + rowT[row]] : xyzr[rowO[row] * 4 + (jj - n)];
            gradx[(size_t)(dbase + row) * d + jj] = pe_chain_rule(gf[row * 33 + jj], gf[row * 33 + d + jj], gf[row * 33 + 2 * d + jj], x);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The hidden part of the pass-2 backward on 4-ROW GROUPS (gemm4): for workgroups whose rows do not fill a 16-row tile.  At
// N = 1024, k = 5 a CU's share is 4 rollouts = 20 rows: five groups of 4 where the 16-row shape needs two blocks (32 rows of
// matrix-pipe time) -- 10240 instead of 16384 MFMA cycles per SIMD and layer.  ReLU networks without skip concatenations;
// the same fmaf chains in the same k order as pass2_backward<ACT, 16 | 32>, so the same bits.
//   waves 0-3: columns 64 w .. +63 of ALL row groups (one wave per SIMD: a second one would fetch the same weight fragments from L2 a second time; weights through the 8-chunk ring across the layers); thread: column 64 w + lane.
// sm.maskG4[l * 256 + column]: bit e = ReLU mask of element (row e, column) at hidden level l (written by the caller BEFORE
// this call, which overwrites the tile buffer).  Leaves the gradient at the first layer's pre-activations in sm.Hs
// (32 rows, the rest 0).
// ------------------------------------------------------------------------------------------------
template <int NG>
__device__ __forceinline__ void pass2_backward_hidden_g4(const MlpDev& m, const P2Smem& sm) {
    float* Hs = sm.Hs;
    const int* rowMin = sm.rowMin;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 3, col = 64 * cw + lane, pcol = omds_kpos(col);   // the column's position in the k-permuted tile
    const bool mine = wave < 4;   // waves 4-7 take no part in the GEMMs: a second wave per SIMD would fetch the same weight fragments
                                  // from L2 a second time (0.5 B per FLOP on this shape), and that, not the matrix pipe, then bounds the layer
    constexpr int PD = 8;
    W4Ring<PD> ring;
    if (mine) {   // the first chunks of the first GEMM are on their way during the seed
        ring.bind(m.Wb4, m.nhh, cw, lane);
        ring.fill(m.nhh - 1);
    }
    // rows past the last group are zero for the first-layer backward, which runs on the whole 32-row tile
    for (int e = tid; e < (32 - 4 * NG) * OMDS_WIDTH; e += P2_NT) Hs[(4 * NG + e / OMDS_WIDTH) * LDH + (e % OMDS_WIDTH)] = 0.f;
    // ---- seed: dy[minIdx]/dH_last = Wlast[minIdx], masked by the last hidden layer
    if (mine) {
        const uint32_t bseed = sm.maskG4[m.nhh * 256 + (tid & 255)];
#pragma unroll
        for (int e = 0; e < 4 * NG; ++e) {
            const float gz = m.Wlraw[(size_t)rowMin[e] * OMDS_WIDTH + col];
            Hs[e * LDH + pcol] = ((bseed >> e) & 1u) ? gz : 0.f;
        }
    }
    __syncthreads();
    OMDS_TL_STAMP(4);
    for (int l = m.nhh - 1; l >= 0; --l) {
        f32x4 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (mine) gemm4<NG, PD>(Hs, ring, l, l - 1, lane, acc);
        if (l == m.nhh - 1) OMDS_TL_STAMP(5);   // the first GEMM, this wave's share
        __syncthreads();   // every wave has finished reading the tile
        if (l == m.nhh - 1) OMDS_TL_STAMP(6);
        if (mine) {
            const uint32_t bl = sm.maskG4[l * 256 + (tid & 255)];
#pragma unroll
            for (int e = 0; e < 4 * NG; ++e) Hs[e * LDH + pcol] = ((bl >> e) & 1u) ? acc[e >> 2][e & 3] : 0.f;
        }
        __syncthreads();
        if (l == m.nhh - 1) OMDS_TL_STAMP(7);
    }
    OMDS_TL_STAMP(8);
}

// ------------------------------------------------------------------------------------------------
// Pass 2 -- forward AND backward -- on 4-ROW GROUPS, for the all-fp32 tail of large batches (k_tail<ND, ACT, 4>): a workgroup
// owns 4 NG = 20 network rows (four rollouts at k = 5), so that 1024 rollouts are 256 workgroups, one per CU, each multiplying
// five groups of 4 rows per hidden layer where the 32-row tile of 171 workgroups multiplies 32 (11 264 against 16 384 matrix-
// pipe cycles per SIMD and layer, forward and backward).  ReLU networks without skip concatenations.
//   layer 1 (K = 32): the 32-row product of pass2_body on all eight waves (rows >= 4 NG are padding);
//   hidden layers: gemm4 on waves 0-3 (columns 64 w .. +63), forward weights MlpDev::Wf4 through the ring across the layers;
//     the epilogue (bias last, ReLU) leaves the level in the tile and its masks as sm.maskG4[(l + 1) * 256 + column], bit e =
//     row e -- the layout pass2_backward_hidden_g4 reads; level 0's word is read back from the tile (h > 0 <=> z > 0);
//   last layer, arg-min link, distance: as pass2_body; then pass2_backward_hidden_g4 + p2_backward_first<32>.
// The same fmaf chains in the same k order as pass2_body<ACT, 32 | 16>: the same bits per row (tests/test_gpu_screen.py runs the
// shapes against each other).
// ------------------------------------------------------------------------------------------------
// What pass2_body_g4 needs that does not depend on WHICH obstacles the rows are: requested before the caller's top-k, so that the
// kernel's first dependent round trips (a fetch from another XCD's writes takes microseconds) run side by side instead of in a row
struct P2G4Pre {
    W4Ring<8> ring;      // waves 0-3: the first chunks of the first hidden product
    float b1v, bnext;    // biases of layer 1 (this thread's column of the 32-row product) and of the first hidden product
    uint32_t q[2];       // the rollout halves of this thread's two encoded-input elements: rows (tid >> 5) and 16 + (tid >> 5), feature tid & 31
};
// t_of_row(r) = the rollout of tile row r, or -1 (what the top-k will write to sm.rowT[r])
template <typename TOfRow>
__device__ __forceinline__ void pass2_g4_prefetch(const MlpDev& m, const float* __restrict__ Fq, TOfRow t_of_row, P2G4Pre& pre) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 3, col4 = 64 * cw + lane;
    if (wave < 4) {
        pre.ring.bind(m.Wf4, m.nhh, cw, lane);
        pre.ring.fill(0);
    }
    pre.b1v = m.b1[P2Geo<32>::col(0, wave, lane)];
    pre.bnext = wave < 4 ? m.bh[col4] : 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = t_of_row((tid >> 5) + 16 * h);
        pre.q[h] = t >= 0 ? __builtin_bit_cast(uint32_t, Fq[(size_t)t * OMDS_FROW + (tid & 31)]) : 0u;
    }
}

template <int NG>
__device__ __forceinline__ void pass2_body_g4(const MlpDev& m, P2Smem& sm, P2G4Pre& pre, const float* __restrict__ Fp,
                                              const float* __restrict__ radius, const float* __restrict__ xyzr, int R0, int total_rows,
                                              const float* __restrict__ qT, int ldq, float* gradx, float* drow, int dbase, int dbg = 0) {
    using G = P2Geo<32>;
    float* Hs = sm.Hs;
    int* rowT = sm.rowT;
    int* rowO = sm.rowO;
    int* rowMin = sm.rowMin;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 3, col4 = 64 * cw + lane, pcol = omds_kpos(col4);
    const bool mine = wave < 4;
    sm.maskG4 = reinterpret_cast<uint32_t*>(sm.maskL);   // (nhh + 1) * 256 words <= the (nhh + 1) * 512 halfwords of maskL
    constexpr int PD = 8;
    W4Ring<PD>& ring = pre.ring;
    const float b1v = pre.b1v;
    float bnext = pre.bnext;
    // ---- the rows' encoded inputs at positions 0..31: Fq[t] | Fp[o] (padding rows: zero), as pass2_body; the rollout half is
    //      already here, the two obstacle halves are one round trip -----------------------------------------------------------
    {
        uint32_t pv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = (tid >> 5) + 16 * h;
            pv[h] = rowT[r] >= 0 ? __builtin_bit_cast(uint32_t, Fp[(size_t)rowO[r] * OMDS_FROW + (tid & 31)]) : 0u;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            reinterpret_cast<uint32_t*>(Hs)[((tid >> 5) + 16 * h) * LDH + omds_kpos(tid & 31)] = rowT[(tid >> 5) + 16 * h] >= 0 ? (pre.q[h] | pv[h]) : 0u;
    }
    __syncthreads();
    OMDS_TL_STAMP(3);
    if (dbg == 16) return;
    // ---- layer 1 on the 32-row tile ---------------------------------------------------------------------------------------
    {
        float acc[G::NV];
        p2_gemm<32>(Hs, m, -1, false, wave, lane, acc);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < G::NV; ++r) Hs[G::row(r, lane) * LDH + G::pos(r, wave, lane)] = actf(acc[r] + b1v, OMDS_ACT_RELU);
    }
    __syncthreads();
    OMDS_TL_STAMP(17);
    if (dbg == 10) return;
    if (mine) {
        uint32_t bits = 0;
#pragma unroll
        for (int e = 0; e < 4 * NG; ++e) bits |= (Hs[e * LDH + pcol] > 0.f ? 1u : 0u) << e;
        sm.maskG4[col4] = bits;
    }
    // ---- hidden -> hidden layers on 4-row groups --------------------------------------------------------------------------
    for (int l = 0; l < m.nhh; ++l) {
        f32x4 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float bv = bnext;
        if (mine && l + 1 < m.nhh) bnext = m.bh[(l + 1) * OMDS_WIDTH + col4];
        if (mine) gemm4<NG, PD>(Hs, ring, l, l + 1 < m.nhh ? l + 1 : -1, lane, acc);
        __syncthreads();   // every wave has finished reading the tile
        if (mine) {
            uint32_t bits = 0;
#pragma unroll
            for (int e = 0; e < 4 * NG; ++e) {
                const float z = acc[e >> 2][e & 3] + bv;
                bits |= (z > 0.f ? 1u : 0u) << e;
                Hs[e * LDH + pcol] = actf(z, OMDS_ACT_RELU);
            }
            sm.maskG4[(l + 1) * 256 + col4] = bits;
        }
        __syncthreads();
    }
    OMDS_TL_STAMP(18);
    if (dbg == 11) return;
    // ---- last layer, arg-min over ALL raw outputs (robot_sdf.py:155), distance of that link: pass2_body's ---------------------
    p2_last_layer<32>(m, Hs, rowMin, rowO, radius, R0, total_rows, drow, dbase, nullptr, nullptr, nullptr, 0u, -1);
    __syncthreads();
    if (dbg == 12) return;
    pass2_backward_hidden_g4<NG>(m, sm);
    if (dbg == 14) return;
    p2_backward_first<32>(m, sm, xyzr, R0, total_rows, qT, ldq, gradx, dbase, dbg);
}

// SDF training on the device (gfx950): what mlp_learn/train_sdf.py:96-151 does per epoch -- a FULL-BATCH forward of the
// distance network (network_macros_mod.py:137-146: [x, sin x, cos x] -> Linear + act ... -> Linear), F.mse_loss(reduction =
// 'mean'), backward through every layer and one torch.optim.Adam step -- as exact-fp32 MFMA GEMMs.  The scheduler
// (ReduceLROnPlateau), the validation metrics and the checkpoint dictionary stay on the host (tools/train_sdf_hip.py): they are
// O(1) per epoch.
//
// One GEMM kernel serves the three shapes of a layer (B = batch rows, the large dimension):
//   forward            Z  [B x out] = H [B x in]   . W^T            (W is [out x in], row-major like torch)
//   input gradient     Gi [B x in]  = G [B x out]  . W
//   weight gradient    dW [out x in] = G^T         . H              (contraction over B: split over workgroups, the partial
//                                                                    products summed in a fixed order -- deterministic)
// 64 x 64 output tiles, 4 waves, v_mfma_f32_32x32x2_f32 (an fmaf chain in k order, tools/ubench/mfma_order.hip), operands staged
// through LDS 16 k at a time with either operand read transposed.  Everything else of a step is elementwise or a column sum.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "mlp_device.h"   // gemm256: the fp32 MFMA GEMM core of the rollout kernels (k_pass1 runs it at 0.93 of the peak)

namespace {

constexpr int TM = 128, TN = 128, TK = 16;

// C[z] (M x N, ldc) = opA(A) (M x K) . opB(B) (K x N) over k in [z * kchunk, min(K, (z + 1) * kchunk)).
// TA: A is stored [K][M] (lda = row stride of the stored matrix); else [M][K].  TB: B is stored [N][K]; else [K][N].
// 128 x 128 output tile per workgroup, 4 waves in 2 x 2, each wave 64 x 64 as 2 x 2 MFMA tiles: four MFMAs per four LDS
// operand reads (a 64 x 64 workgroup tile with one MFMA per two reads ran 28 TFLOP/s on the training shapes).
// EPI 0: C = product.  EPI 1 (forward): C = act(product + bias[n]) (act < 0: none).  EPI 2 (input gradient): C = product * act'(z)
// read from the layer's stored activation aux[m][n] = act(z) -- the elementwise passes over [B x width] fused into the stores.
template <bool TA, bool TB, int EPI>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                              float* __restrict__ C, int ldc, int M, int N, int K, int kchunk, size_t cstride,
                                              const float* __restrict__ aux, int act) {
    // [k][m] / [k][n] tiles, row stride 132 floats: the MFMA operand reads (32 consecutive m of row k, lanes 32-63 row k + 1) are
    // conflict-free in their lane groups, and so are the transposing stores of a k-contiguous operand (bank = 16 kq + m below)
    constexpr int LDT = TM + 4;
    __shared__ __attribute__((aligned(16))) float As[TK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[TK][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;   // the batch dimension on grid.x: its limit is 2^31 - 1 tiles, grid.y stops at 65535
    const int k_begin = blockIdx.z * kchunk, k_end = min(K, k_begin + kchunk);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // The two operand tiles of a k step (128 x 16 elements each, 8 per thread; zeros outside the matrices) travel global -> registers
    // -> LDS.  The NEXT step's global loads are issued before this step's MFMAs and parked in registers, so their latency hides
    // under 32 MFMAs per wave instead of standing between two barriers (round 4: the trainer's three GEMM shapes 64-71 -> see
    // EXPERIMENTS.md C 4.8; the arithmetic of an output element is unchanged: the same products in the same k order).
    // Staging in 16-byte pieces (two per thread and operand): an operand stored with k contiguous (A of the forward and the input
    // gradient, B = W of the forward) is read as float4 along k -- thread (row = idx >> 2, k quad = idx & 3) -- and transposed into the
    // [k][row] tile by four scalar stores; an operand stored with its row index contiguous (G^T, H of the weight gradient, W of the
    // input gradient) is read as float4 along the row and stored as one.  Pieces that straddle an edge, or operands whose row
    // stride is not a multiple of four floats (the 3 d = 15 / 30 input features), fall back to guarded scalar loads.
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 ra[2], rb[2];
    const bool a_vec = (lda & 3) == 0 && (reinterpret_cast<size_t>(A) & 15) == 0, b_vec = (ldb & 3) == 0 && (reinterpret_cast<size_t>(B) & 15) == 0;
    auto load_op = [&](const float* __restrict__ P, int ld, int r0, int R, bool kcontig, bool vec, int k0, f4 (&r)[2]) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = tid + 256 * e;
            if (kcontig) {
                const int kq = idx & 3, row = idx >> 2, g = r0 + row, gk = k0 + 4 * kq;
                if (vec && g < R && gk + 3 < k_end) {
                    r[e] = *reinterpret_cast<const f4*>(P + (size_t)g * ld + gk);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[e][j] = (g < R && gk + j < k_end) ? P[(size_t)g * ld + gk + j] : 0.f;
                }
            } else {
                const int rq = idx & 31, kk = idx >> 5, g = r0 + 4 * rq, gk = k0 + kk;
                if (vec && gk < k_end && g + 3 < R) {
                    r[e] = *reinterpret_cast<const f4*>(P + (size_t)gk * ld + g);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[e][j] = (gk < k_end && g + j < R) ? P[(size_t)gk * ld + g + j] : 0.f;
                }
            }
        }
    };
    auto store_op = [&](float (&T)[TK][LDT], bool kcontig, const f4 (&r)[2]) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = tid + 256 * e;
            if (kcontig) {
                const int kq = idx & 3, row = idx >> 2;
#pragma unroll
                for (int j = 0; j < 4; ++j) T[4 * kq + j][row] = r[e][j];
            } else {
                const int rq = idx & 31, kk = idx >> 5;
                *reinterpret_cast<f4*>(&T[kk][4 * rq]) = r[e];
            }
        }
    };
    auto load_tiles = [&](int k0) {
        load_op(A, lda, m0, M, !TA, a_vec, k0, ra);
        load_op(B, ldb, n0, N, TB, b_vec, k0, rb);
    };
    auto store_tiles = [&]() {
        store_op(As, !TA, ra);
        store_op(Bs, TB, rb);
    };
    if (k_begin < k_end) {
        load_tiles(k_begin);
        store_tiles();
    }
    __syncthreads();
    for (int k0 = k_begin; k0 < k_end; k0 += TK) {
        const bool more = k0 + TK < k_end;
        if (more) load_tiles(k0 + TK);      // in flight across the MFMAs below
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            const float a0 = As[kk + (lane >> 5)][wm + (lane & 31)], a1 = As[kk + (lane >> 5)][wm + 32 + (lane & 31)];
            const float b0 = Bs[kk + (lane >> 5)][wn + (lane & 31)], b1 = Bs[kk + (lane >> 5)][wn + 32 + (lane & 31)];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();                     // every wave has read this step's tiles
        if (more) store_tiles();
        __syncthreads();
    }
    float* Cz = C + (size_t)blockIdx.z * cstride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {   // C/D layout: lane l, register r -> row (r & 3) + 8 (r >> 2) + 4 (l >> 5), col l & 31
                const int gm = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), gn = n0 + wn + 32 * j + (lane & 31);
                if (gm < M && gn < N) {
                    float v = acc[i][j][r];
                    if constexpr (EPI == 1) {
                        v += aux[gn];
                        v = act == 0 ? fmaxf(v, 0.f) : (act == 1 ? tanhf(v) : v);
                    } else if constexpr (EPI == 2) {
                        const float h = aux[(size_t)gm * ldc + gn];
                        v = act == 0 ? (h > 0.f ? v : 0.f) : v * (1.f - h * h);
                    }
                    Cz[(size_t)gm * ldc + gn] = v;
                }
            }
}

// ---- the tall GEMMs of a 256-wide layer: C [M x N] = A [M x K] . B with M = the batch, K = 256 and 128 < N <= 256 ------------------------
// (forward H . W^T and input gradient G . W of the hidden layers: 2/3 of an epoch's FLOPs).  The general kernel above stages both
// operands through LDS 16 k at a time, two barriers per step: 0.62 of the fp32 MFMA peak.  Here a workgroup owns 64 rows and ALL
// columns: its A tile [64 x 256] sits in LDS for the whole product (row stride 260: conflict-free 16-byte fragment reads), the
// weights arrive as MFMA B fragments straight from L2 into registers (k_pack256 below re-packs the layer's matrix once per call:
// 256 KB, every workgroup streams the same bytes), and the product is gemm256 of mlp_device.h -- fully unrolled, no VALU in the
// loop, loads pinned between the MFMAs.  Two workgroups per CU: one's tile load / epilogue runs beside the other's GEMM.
// Arithmetic of an output element: ONE fmaf chain over ascending k, like the general kernel's; deterministic, independent of M.
__global__ __launch_bounds__(256) void k_pack256(const float* __restrict__ W, int ld, int N, int K, int trans, float4* __restrict__ P) {
    const int idx = blockIdx.x * 256 + threadIdx.x;       // [8 column blocks][32 k chunks][64 lanes]
    const int lane = idx & 63, c = (idx >> 6) & 31, cb = idx >> 11;
    const int n = 32 * cb + (lane & 31), k0 = 8 * c + (lane >> 5);     // lane half h holds k = 8c + h, + 2, + 4, + 6: MFMA step m contracts k = 8c + 2m, + 1
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + 2 * j;
        v[j] = (n < N && k < K) ? (trans ? W[(size_t)k * ld + n] : W[(size_t)n * ld + k]) : 0.f;   // B(k, n): W[n][k] (forward) or W[k][n] (input gradient)
    }
    P[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

constexpr int TALL_ROWS = 64;
constexpr int TALL_BUF = TALL_ROWS * LDH * 4;        // bytes of one LDS tile [64][LDH]
// LDS-DMA (screen_kernel.hip's idiom): 4 bytes per lane from (wave-uniform base + per-lane byte offset) to LDS [lds_dst + 4 * lane]
// (256 B per instruction); the immediate advances the global AND the LDS address.  The lane picks WHICH element of the row lands in
// its LDS slot, i.e. a row can arrive k-permuted.  Invisible to hipcc's s_waitcnt bookkeeping by design: completion is waited for
// explicitly (tall_wait_all).  nt: streamed once -- the weights and the next rows keep the L2.
template <int IMM>
__device__ __forceinline__ void tall_dma4(const void* gbase_uniform, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:%4 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(gbase_uniform), "s"(lds_dst), "n"(IMM)
                 : "memory");
}
// lgkmcnt(0) too: the barrier also orders plain LDS stores (the parked outputs of the previous tile, the zero fill of rows past M)
// that other waves read behind it, and hipcc inserts no wait in front of an inline-asm barrier
__device__ __forceinline__ void tall_wait_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// PERSISTENT workgroups (one per CU, 8 waves), two LDS tiles, and the WEIGHTS IN REGISTERS.  A wave owns 32 output columns for the
// whole launch, so its 32 k-chunks of B fragments (one float4 each: 128 VGPRs) are tile-invariant: they are fetched once, and the
// product loop of a tile contains no vector-memory instruction at all -- A fragments come from LDS, MFMAs, nothing else.  That is
// what lets the HBM traffic overlap the product: vmcnt is an in-order counter per wave, so a wave that prefetches the next tile
// and then waits for its next weight fragment (an L2 hit) waits for the prefetch too.  Every form of this kernel that streamed
// its weights ran the traffic in series with the MFMAs and stayed at 0.60-0.65 of the peak (two workgroups per CU; persistent +
// double-buffered; separate mover waves, which an MFMA stream starves of issue slots: EXPERIMENTS.md; tools/studies/tall_ablation.sh:
// the product alone 1.00 ms, with its traffic 1.33-1.53).  Per tile: the rows of tile t + 1 arrive in the other buffer by LDS-DMA
// (8 instructions per wave, issued before the product), the parked output of tile t - 1 leaves for C (8 x (ds_read_b128 + 16-byte
// store) per thread, masked with the stored activation for the input gradient, which is requested a tile ahead), and ONE wait for
// all of it stands at the top of the next tile, a whole product later.
// Requires K == lda == 256 (a row is exactly one 1 KB DMA piece) and a 16-byte aligned A; N <= 256 arbitrary.
template <int EPI>
__global__ __launch_bounds__(512, 1) void k_gemm_tall(const float* __restrict__ A, const float4* __restrict__ P, float* __restrict__ C,
                                                      int ldc, int M, int N, const float* __restrict__ aux, int act, int ntiles, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float Hs[];     // [2][64][LDH]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(Hs);
    const int kq = 4 * lane;
    const bool c_vec = (ldc & 3) == 0 && (reinterpret_cast<size_t>(C) & 15) == 0 && (EPI != 2 || (reinterpret_cast<size_t>(aux) & 15) == 0);
    // One row per call: a wave moves row p * 8 + wave of a tile -- the SAME rows in and out of a buffer, so a buffer's turn from parked
    // output to next input needs no workgroup barrier, only this wave's own program order (read the row out, then DMA over it).
    // The A rows arrive k-PERMUTED inside every group of 8: LDS slot 8c + 4h + j holds k = 8c + 2j + h, so that the one 16-byte
    // fragment read of lane half h is k = 8c + h + {0, 2, 4, 6} -- MFMA step m then contracts k = 8c + 2m and 8c + 2m + 1: ASCENDING k,
    // the fmaf chain of the general kernel above and of a plain sgemm micro-kernel (see k ORDER below).  The permutation costs nothing
    // on the way in: LDS-DMA at 4 bytes per lane lets each lane name the element it fetches (4 instructions per row instead of 1).
    const unsigned perm_voff = 4u * (unsigned)(8 * (lane >> 3) + 2 * (lane & 3) + ((lane >> 2) & 1));   // slot `lane` of a 64-slot piece <- this k (bytes)
    auto fetch_half = [&](int tile, int buf, int p, int half) {   // half a row (2 pieces of 256 B) of tile -> LDS buffer buf; rows past M are zero-filled
        if (OMDS_DBG(dbg) & 1) return;                    // experiment builds: no A traffic (the product runs on whatever the buffer holds)
        const int row = p * 8 + wave;
        const size_t g = (size_t)tile * TALL_ROWS + row;
        const unsigned dst = lds0 + (unsigned)(buf * TALL_BUF + row * (LDH * 4));
        if (g < (size_t)M) {
            if (half == 0) { tall_dma4<0>(A + g * 256, perm_voff, dst); tall_dma4<256>(A + g * 256, perm_voff, dst); }
            else { tall_dma4<512>(A + g * 256, perm_voff, dst); tall_dma4<768>(A + g * 256, perm_voff, dst); }
        } else if (lane < 32) {
            *reinterpret_cast<float4*>(Hs + buf * (TALL_ROWS * LDH) + row * LDH + 128 * half + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto aux_row = [&](int tile, int p) {                 // the stored activation the input gradient of `tile` is masked with
        const size_t g = (size_t)tile * TALL_ROWS + p * 8 + wave;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g < (size_t)M && !(OMDS_DBG(dbg) & 4)) {
            const float* src = aux + g * ldc + kq;
            if (c_vec && kq + 3 < N) {
                typedef float nt_f4 __attribute__((ext_vector_type(4)));
                const nt_f4 o = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(src));
                v = make_float4(o.x, o.y, o.z, o.w);
            } else {
                if (kq < N) v.x = src[0];
                if (kq + 1 < N) v.y = src[1];
                if (kq + 2 < N) v.z = src[2];
                if (kq + 3 < N) v.w = src[3];
            }
        }
        return v;
    };
    auto park_read = [&](int buf, int p) {                // parked output row p * 8 + wave of buffer buf
        return *reinterpret_cast<const float4*>(Hs + buf * (TALL_ROWS * LDH) + (p * 8 + wave) * LDH + kq);
    };
    auto drain_row = [&](int tile, int p, float4 v, const float4& a) {   // parked output row of `tile` (already read: v) -> C
        if (OMDS_DBG(dbg) & 2) return;                    // experiment builds: no C traffic
        const size_t g = (size_t)tile * TALL_ROWS + p * 8 + wave;
        if (g >= (size_t)M) return;                       // wave-uniform
        if constexpr (EPI == 2) {
            if (act == 0) {
                v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
            } else {
                v.x *= 1.f - a.x * a.x; v.y *= 1.f - a.y * a.y; v.z *= 1.f - a.z * a.z; v.w *= 1.f - a.w * a.w;
            }
        }
        float* dst = C + g * ldc + kq;
        if (c_vec && kq + 3 < N) {                        // streamed once: non-temporal, so that the weights and the next rows keep the L2
            typedef float nt_f4 __attribute__((ext_vector_type(4)));
            nt_f4 o = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(o, reinterpret_cast<nt_f4*>(dst));
        } else {
            if (kq < N) dst[0] = v.x;
            if (kq + 1 < N) dst[1] = v.y;
            if (kq + 2 < N) dst[2] = v.z;
            if (kq + 3 < N) dst[3] = v.w;
        }
    };
    // this wave's B fragments, all 32 k-chunks: lane l's float4 = B(k = 8c + (l >> 5) + {0, 2, 4, 6}, n = 32 wave + (l & 31))  (k_pack256)
    float4 wreg[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) wreg[c] = P[(size_t)wave * (32 * 64) + c * 64 + lane];
    const int col = 32 * wave + (lane & 31);
    float bias = 0.f;
    if constexpr (EPI == 1) bias = col < N ? aux[col] : 0.f;
    float4 hprev[TALL_ROWS / 8], hcur[TALL_ROWS / 8];
#pragma unroll
    for (int p = 0; p < TALL_ROWS / 8; ++p) hprev[p] = hcur[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    int tile = blockIdx.x, prev = -1, it = 0;
    if (tile < ntiles) {
#pragma unroll
        for (int p = 0; p < TALL_ROWS / 8; ++p) { fetch_half(tile, 0, p, 0); fetch_half(tile, 0, p, 1); }
    }
    for (; tile < ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        tall_wait_all();                             // this tile's rows have landed, the previous tile's output is parked (and everything older has retired)
        const int next = tile + gridDim.x;
        f32x16 acc0, acc1;
        float4 rd[TALL_ROWS / 8];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
        // The product, with the tile's memory traffic dealt into it one instruction per k-chunk (a CU has ONE vector-memory path:
        // issued as a block in front of the product, the 24 instructions per wave kept the matrix pipe waiting 1.3-2.4 us per tile):
        // chunks 0-8 send the parked rows of tile t - 1 to C (the LDS read of a row one chunk ahead of its store), chunks 9-24 request
        // the rows of tile t + 1 over them (half a row each), chunks 25-31 request this tile's mask rows.  Nothing in here waits on vmcnt.
        // k ORDER: ascending -- MFMA step m of chunk c contracts k = 8c + 2m (lanes 0-31) and 8c + 2m + 1 (lanes 32-63): torch-CPU's sgemm
        // order too, so a run of this trainer from trained ReLU weights tracks the torch run to 1e-6 in the loss over 30 epochs --
        // near-dead units get the SAME rounding-level gradients, and Adam's first steps are lr * sign(g) whatever |g| is; in the
        // position order 8c + {0, 4, 1, 5, ...} 5 % of the weights had taken a step the other way after 5 epochs (tests/test_gpu_train.py).
        const float* arow = Hs + buf * (TALL_ROWS * LDH) + (lane & 31) * LDH + 4 * (lane >> 5);
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const float4 a0 = *reinterpret_cast<const float4*>(arow + 8 * c);
            const float4 a1 = *reinterpret_cast<const float4*>(arow + 32 * LDH + 8 * c);
            const float4 w = wreg[c];
            if (c < 8) { if (prev >= 0) rd[c] = park_read(buf ^ 1, c); }
            if (c >= 1 && c <= 8) { if (prev >= 0) drain_row(prev, c - 1, rd[c - 1], hprev[c - 1]); }
            else if (c >= 9 && c <= 24) { if (next < ntiles) fetch_half(next, buf ^ 1, (c - 9) >> 1, (c - 9) & 1); }
            else if (c >= 25) {
                if constexpr (EPI == 2) {
                    hcur[c - 25] = aux_row(tile, c - 25);
                    if (c == 31) hcur[7] = aux_row(tile, 7);
                }
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, w.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, w.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, w.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, w.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, w.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, w.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, w.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, w.w, acc1, 0, 0, 0);
        }
        __syncthreads();                             // every wave has read the A tile: it becomes the parked output
        if constexpr (EPI == 1) {                    // the activation switch outside the element loop (it was a branch per element)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] += bias; acc1[r] += bias; }
            if (act == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] = fmaxf(acc0[r], 0.f); acc1[r] = fmaxf(acc1[r], 0.f); }
            } else if (act == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] = tanhf(acc0[r]); acc1[r] = tanhf(acc1[r]); }
            }
        }
        float* park = Hs + buf * (TALL_ROWS * LDH) + (4 * (lane >> 5)) * LDH + col;   // crow(r, lane) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            park[((r & 3) + 8 * (r >> 2)) * LDH] = acc0[r];
            park[(32 + (r & 3) + 8 * (r >> 2)) * LDH] = acc1[r];
        }
        prev = tile;
        if constexpr (EPI == 2) {
#pragma unroll
            for (int p = 0; p < TALL_ROWS / 8; ++p) hprev[p] = hcur[p];
        }
    }
    __syncthreads();
    if (prev >= 0) {
#pragma unroll
        for (int p = 0; p < TALL_ROWS / 8; ++p) drain_row(prev, p, park_read((it - 1) & 1, p), hprev[p]);
    }
}

// ---- the thin products at the two ends of the network: C [M x N] = A [M x K] . B with K <= 32 (the encoded input, 3 d = 12 / 30 wide, into the
// first hidden layer; the output gradient, C = 2 / 9 wide, back into the last hidden layer) ----------------------------------------------------
// Pure streaming: 1 GB of output (and, for the input gradient, 1 GB of mask) per million rows against 6 GFLOP.  The general kernel moves its
// 128 x 128 tiles through the MFMA machinery for one or two k steps and reaches 1.7-2.3 TB/s; here a thread owns one output COLUMN, keeps that
// column's K weights in registers, and walks down the rows of its workgroup's chunk: the A rows of a chunk sit in LDS (every lane reads the same
// address: a broadcast), one fmaf per k in ASCENDING k from a zero accumulator -- the chain of the MFMA kernels, bit for bit -- then the same
// epilogue (EPI 1: + bias, activation; EPI 2: x act'(stored activation)), one coalesced 1 KB store per row and wave quartet.
constexpr int THIN_KMAX = 32, THIN_ROWS = 64;
template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_thin(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, int trans,
                                                   float* __restrict__ C, int ldc, int M, int N, int K, const float* __restrict__ aux, int act) {
    __shared__ __attribute__((aligned(16))) float As[THIN_ROWS][THIN_KMAX];
    const int n = threadIdx.x;                       // this thread's output column
    const bool live = n < N;
    float w[THIN_KMAX];
#pragma unroll
    for (int k = 0; k < THIN_KMAX; ++k) w[k] = (live && k < K) ? (trans ? W[(size_t)k * ldw + n] : W[(size_t)n * ldw + k]) : 0.f;   // B(k, n)
    float bias = 0.f;
    if constexpr (EPI == 1) bias = live ? aux[n] : 0.f;
    const int K4 = (K + 3) & ~3;
    for (size_t m0 = (size_t)blockIdx.x * THIN_ROWS; m0 < (size_t)M; m0 += (size_t)gridDim.x * THIN_ROWS) {
        __syncthreads();                             // the previous chunk has been consumed
        for (int i = threadIdx.x; i < THIN_ROWS * K4; i += 256) {
            const int r = i / K4, k = i - r * K4;
            As[r][k] = (m0 + r < (size_t)M && k < K) ? A[(m0 + r) * lda + k] : 0.f;
        }
        __syncthreads();
        const int rows = (int)((size_t)M - m0 < THIN_ROWS ? (size_t)M - m0 : THIN_ROWS);
        for (int r = 0; r < rows; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < THIN_KMAX / 4; ++k4) {
                if (4 * k4 < K4) {                   // uniform
                    const float4 a = *reinterpret_cast<const float4*>(&As[r][4 * k4]);
                    acc = fmaf(a.x, w[4 * k4], acc);
                    acc = fmaf(a.y, w[4 * k4 + 1], acc);
                    acc = fmaf(a.z, w[4 * k4 + 2], acc);
                    acc = fmaf(a.w, w[4 * k4 + 3], acc);
                }
            }
            if (live) {
                float v = acc;
                const size_t o = (m0 + r) * ldc + n;
                if constexpr (EPI == 1) {
                    v += bias;
                    v = act == 0 ? fmaxf(v, 0.f) : (act == 1 ? tanhf(v) : v);
                } else if constexpr (EPI == 2) {
                    const float h = __builtin_nontemporal_load(aux + o);
                    v = act == 0 ? (h > 0.f ? v : 0.f) : v * (1.f - h * h);
                }
                __builtin_nontemporal_store(v, C + o);
            }
        }
    }
}

// The thin-OUTPUT forward (the last layer, 256 -> C = 2 / 9): out [M x N] = A [M x K] . W^T + bias with N <= 16, K <= 256.  Streams A once (1 GB
// per million rows; the general kernel reads it at 1.7 TB/s through 128-column tiles of which 2 columns are real).  A chunk of 32 rows sits in LDS
// (row stride 260: conflict-free 16-byte reads down the rows), the whole W beside it; thread (row, group g) computes outputs g, g + 4, ... of its
// row as fmaf chains in ASCENDING k from zero, then + bias (and the activation, if the layer has one): the general kernel's bits.
constexpr int THINN_ROWS = 32, THINN_NMAX = 16;
__global__ __launch_bounds__(128) void k_gemm_thin_out(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* __restrict__ C,
                                                       int ldc, int M, int N, int K, const float* __restrict__ bias, int act) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* As = sm;                                   // [32][LDH]
    float* Ws = sm + THINN_ROWS * LDH;                // [N][LDH]
    const int tid = threadIdx.x, row = tid & 31, g = tid >> 5;
    const int K4 = (K + 3) & ~3;
    for (int i = tid; i < N * LDH; i += 128) { const int n = i / LDH, k = i - n * LDH; Ws[i] = k < K ? W[(size_t)n * ldw + k] : 0.f; }
    const bool vec = (lda & 3) == 0 && (reinterpret_cast<size_t>(A) & 15) == 0 && (K & 3) == 0;
    for (size_t m0 = (size_t)blockIdx.x * THINN_ROWS; m0 < (size_t)M; m0 += (size_t)gridDim.x * THINN_ROWS) {
        __syncthreads();
        for (int i = tid; i < THINN_ROWS * (K4 / 4); i += 128) {
            const int r = i / (K4 / 4), k = 4 * (i - r * (K4 / 4));
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + r < (size_t)M) {
                const float* src = A + (m0 + r) * lda + k;
                if (vec) {
                    typedef float nt_f4 __attribute__((ext_vector_type(4)));
                    const nt_f4 o = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(src));
                    v = make_float4(o.x, o.y, o.z, o.w);
                } else {
                    if (k < K) v.x = src[0];
                    if (k + 1 < K) v.y = src[1];
                    if (k + 2 < K) v.z = src[2];
                    if (k + 3 < K) v.w = src[3];
                }
            }
            *reinterpret_cast<float4*>(As + r * LDH + k) = v;
        }
        __syncthreads();
        if (m0 + row >= (size_t)M) continue;
        float acc[THINN_NMAX / 4];
#pragma unroll
        for (int u = 0; u < THINN_NMAX / 4; ++u) acc[u] = 0.f;
        for (int k = 0; k < K4; k += 4) {
            const float4 a = *reinterpret_cast<const float4*>(As + row * LDH + k);
#pragma unroll
            for (int u = 0; u < THINN_NMAX / 4; ++u) {
                const int n = g + 4 * u;
                if (n < N) {
                    const float4 w = *reinterpret_cast<const float4*>(Ws + n * LDH + k);
                    acc[u] = fmaf(a.x, w.x, acc[u]);
                    acc[u] = fmaf(a.y, w.y, acc[u]);
                    acc[u] = fmaf(a.z, w.z, acc[u]);
                    acc[u] = fmaf(a.w, w.w, acc[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < THINN_NMAX / 4; ++u) {
            const int n = g + 4 * u;
            if (n < N) {
                float v = acc[u] + bias[n];
                v = act == 0 ? fmaxf(v, 0.f) : (act == 1 ? tanhf(v) : v);
                C[(m0 + row) * ldc + n] = v;
            }
        }
    }
}

// The thin weight gradients (the first layer's dW [256 x 3 d], the last layer's dW [C x 256]): one of the two matrices of dW = G^T . H is at most 32
// wide.  The contraction runs over the batch in the SAME chunks as the general kernel's split (blockIdx.y = chunk z), every (z, wide column, thin
// column) sum is one fmaf chain over the chunk's rows in ascending order from zero -- the general kernel's partials bit for bit, summed by the same
// k_sum_partials -- but the wide matrix streams through once, coalesced, WG_BATCH = 32 rows in flight per lane, instead of through MFMA tiles that are 90 %
// padding.  thin_is_n: the thin matrix indexes dW's columns (first layer: Y = H, X = G), else its rows (last layer: Y = G, X = H).
constexpr int WG_TMAX = 32, WG_YROWS = 256, WG_BATCH = 32;
// (256 chunks x 256 columns are 1024 waves, one per SIMD: nothing hides a wave's own load latency but the wave itself -- the next batch of 32 rows
// is requested before the current one is multiplied, and a batch's thin rows are read from LDS one row ahead.)
template <int TT>
__global__ __launch_bounds__(256) void k_wgrad_thin(const float* __restrict__ X, int ldx, int Wd, const float* __restrict__ Y, int ldy, int T,
                                                    int thin_is_n, float* __restrict__ P, int B, int kchunk, size_t cstride, int ldp) {
    __shared__ __attribute__((aligned(16))) float Ys[WG_YROWS][TT];
    const int c = blockIdx.x * 256 + threadIdx.x;    // wide column
    const bool live = c < Wd;
    const int z = blockIdx.y;
    const int r_begin = z * kchunk, r_end = min(B, r_begin + kchunk);
    float acc[TT];
#pragma unroll
    for (int j = 0; j < TT; ++j) acc[j] = 0.f;
    const float* Xc = X + (live ? c : 0);
    auto fetch = [&](float (&x)[WG_BATCH], int row0, int rows_left) {   // rows row0 .. of column c; past the end: zeros (they multiply zeros of Ys)
#pragma unroll
        for (int u = 0; u < WG_BATCH; ++u) x[u] = (live && u < rows_left) ? __builtin_nontemporal_load(Xc + (size_t)(row0 + u) * ldx) : 0.f;
    };
    auto multiply = [&](const float (&x)[WG_BATCH], int yrow0, int rows_left) {
#pragma unroll
        for (int u = 0; u < WG_BATCH; ++u) {
            if (u < rows_left) {                     // uniform
#pragma unroll
                for (int j4 = 0; j4 < TT / 4; ++j4) {
                    const float4 y = *reinterpret_cast<const float4*>(&Ys[yrow0 + u][4 * j4]);
                    acc[4 * j4] = fmaf(x[u], y.x, acc[4 * j4]);
                    acc[4 * j4 + 1] = fmaf(x[u], y.y, acc[4 * j4 + 1]);
                    acc[4 * j4 + 2] = fmaf(x[u], y.z, acc[4 * j4 + 2]);
                    acc[4 * j4 + 3] = fmaf(x[u], y.w, acc[4 * j4 + 3]);
                }
            }
        }
    };
    float xa[WG_BATCH], xb[WG_BATCH];
    fetch(xa, r_begin, r_end - r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += WG_YROWS) {
        __syncthreads();
        for (int i = threadIdx.x; i < WG_YROWS * TT; i += 256) {
            const int r = i / TT, j = i - r * TT;
            Ys[r][j] = (r0 + r < r_end && j < T) ? Y[(size_t)(r0 + r) * ldy + j] : 0.f;
        }
        __syncthreads();
        const int rows = min(WG_YROWS, r_end - r0);
        for (int rb = 0; rb < rows; rb += 2 * WG_BATCH) {        // two batches per trip: xa and xb swap roles without copies
            fetch(xb, r0 + rb + WG_BATCH, r_end - (r0 + rb + WG_BATCH));
            multiply(xa, rb, rows - rb);
            fetch(xa, r0 + rb + 2 * WG_BATCH, r_end - (r0 + rb + 2 * WG_BATCH));
            multiply(xb, rb + WG_BATCH, rows - rb - WG_BATCH);
        }
    }
    if (live) {
        float* Pz = P + (size_t)z * cstride;
#pragma unroll
        for (int j = 0; j < TT; ++j)
            if (j < T) Pz[thin_is_n ? (size_t)c * ldp + j : (size_t)j * ldp + c] = acc[j];
    }
}
static void launch_wgrad_thin(hipStream_t s, int wide_blocks, int splits, const float* X, int ldx, int Wd, const float* Y, int ldy, int T, int thin_is_n,
                              float* P, int B, int kchunk, size_t cstride, int ldp) {
    const dim3 grid(wide_blocks, splits);
    if (T <= 4) hipLaunchKernelGGL(k_wgrad_thin<4>, grid, dim3(256), 0, s, X, ldx, Wd, Y, ldy, T, thin_is_n, P, B, kchunk, cstride, ldp);
    else if (T <= 12) hipLaunchKernelGGL(k_wgrad_thin<12>, grid, dim3(256), 0, s, X, ldx, Wd, Y, ldy, T, thin_is_n, P, B, kchunk, cstride, ldp);
    else if (T <= 16) hipLaunchKernelGGL(k_wgrad_thin<16>, grid, dim3(256), 0, s, X, ldx, Wd, Y, ldy, T, thin_is_n, P, B, kchunk, cstride, ldp);
    else hipLaunchKernelGGL(k_wgrad_thin<32>, grid, dim3(256), 0, s, X, ldx, Wd, Y, ldy, T, thin_is_n, P, B, kchunk, cstride, ldp);
}

__global__ void k_sum_partials(const float* __restrict__ P, int S, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= S; z += 8) {   // eight partials in flight, added in the fixed order z = 0, 1, 2, ...
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = P[(size_t)(z + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < S; ++z) s += P[(size_t)z * n + i];
    out[i] = s;
}

// [x, sin x, cos x] (network_macros_mod.py:139-140)
__global__ void k_encode(const float* __restrict__ x, int B, int d, float* __restrict__ H0) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * d) return;
    const size_t r = i / d;
    const int j = (int)(i - r * d);
    const float v = x[i];
    H0[r * 3 * d + j] = v;
    H0[r * 3 * d + d + j] = omds_sinf(v);
    H0[r * 3 * d + 2 * d + j] = omds_cosf(v);
}

// G = 2 (pred - y) / count  (the gradient of F.mse_loss(..., reduction='mean')); per-block partial sums of (pred - y)^2 in double
__global__ __launch_bounds__(256) void k_mse(const float* __restrict__ pred, const float* __restrict__ y, size_t n, float scale,
                                             float* __restrict__ G, double* __restrict__ partial) {
    __shared__ double red[256];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float dlt = pred[i] - y[i];
        if (G) G[i] = scale * dlt;
        s += (double)dlt * (double)dlt;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// column sums of G [rows x N] (the bias gradient): rows split over blockIdx.y, partials summed in a fixed order afterwards
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ G, size_t rows, int N, size_t rows_per, float* __restrict__ P) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    const size_t r0 = (size_t)blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
    float s = 0.f;
    size_t r = r0;
    for (; r + 8 <= r1; r += 8) {   // eight rows in flight, added in row order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = G[(r + u) * N + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < r1; ++r) s += G[r * N + c];
    P[(size_t)blockIdx.y * N + c] = s;
}

// torch.optim.Adam (default flags: no amsgrad, no weight decay, not capturable), the single-tensor arithmetic:
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  denom = sqrt(v) / sqrt(bc2) + eps;  p -= (lr / bc1) * m / denom
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                       float b1, float b2, float eps, float step_size, float bc2_sqrt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);        // torch: exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;       // torch: exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
}

}  // namespace

struct omds_trainer {
    int dev = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int L = 0, act = 0, d = 0;
    std::vector<int> dims;                 // [L + 1]: 3 d, hidden ..., C
    std::vector<float*> W, b, gW, gb, mW, mb, vW, vb;
    int B = 0, cap = 0;                    // rows of the training set, rows the activation / gradient buffers hold
    float* x = nullptr; float* y = nullptr;
    int Bv = 0, xcap = 0, vcap = 0;        // rows of the validation set (omds_trainer_set_val_data); rows x / y and xv / yv hold
    float* xv = nullptr; float* yv = nullptr;
    std::vector<float*> H;                 // [L + 1] activations: H[0] = encoded input ... H[L] = prediction
    float* G[2] = {nullptr, nullptr};      // gradient ping-pong [cap x max width]
    float4* pack256 = nullptr;             // MFMA fragment pack of the layer a tall GEMM is about to multiply (k_pack256), 256 KB
    float* partial = nullptr;              // split-K partials of the weight gradient / column-sum partials
    size_t partial_floats = 0;
    double* lossp = nullptr; double* h_lossp = nullptr;
    long long step = 0;
};

static thread_local std::string g_train_err;
#define TCK(expr)                                                                          \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) { tr->err = std::string(#expr) + ": " + hipGetErrorString(_e); return OMDS_ERR_HIP; } \
    } while (0)

static void trainer_free_buffers(omds_trainer* tr) {   // activations and gradient ping-pong (sized by the larger data set)
    for (float* p : {tr->G[0], tr->G[1]}) if (p) (void)hipFree(p);
    tr->G[0] = tr->G[1] = nullptr;
    for (float*& p : tr->H) { if (p) (void)hipFree(p); p = nullptr; }
    tr->cap = 0;
}
static void trainer_free_data(omds_trainer* tr) {
    for (float* p : {tr->x, tr->y, tr->xv, tr->yv}) if (p) (void)hipFree(p);
    tr->x = tr->y = tr->xv = tr->yv = nullptr;
    tr->B = tr->Bv = tr->xcap = tr->vcap = 0;
    trainer_free_buffers(tr);
}
static int trainer_reserve(omds_trainer* tr, int rows) {
    if (rows <= tr->cap) return OMDS_OK;
    trainer_free_buffers(tr);
    int wmax = 0;
    for (int v : tr->dims) wmax = std::max(wmax, v);
    for (int i = 0; i <= tr->L; ++i) TCK(hipMalloc(&tr->H[i], (size_t)rows * tr->dims[i] * 4));
    for (int i = 0; i < 2; ++i) TCK(hipMalloc(&tr->G[i], (size_t)rows * wmax * 4));
    tr->cap = rows;
    return OMDS_OK;
}

template <bool TA, bool TB, int EPI>
static void launch_gemm(hipStream_t s, const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K,
                        int splits, int kchunk, size_t cstride, const float* aux = nullptr, int act = -1) {
    const dim3 grid((M + TM - 1) / TM, (N + TN - 1) / TN, splits);   // M may be the batch (millions of rows): grid.x
    hipLaunchKernelGGL((k_gemm<TA, TB, EPI>), grid, dim3(256), 0, s, A, lda, Bm, ldb, C, ldc, M, N, K, kchunk, cstride, aux, act);
}

// The tall fast path: A row-major [M x 256] (K == lda == 256, 16-byte aligned), one output tile row per 64 rows, 128 < N <= 256.  W is the
// layer's [out x in] matrix;
// trans = 0: B(k, n) = W[n][k] (forward, N = out, K = in); trans = 1: B(k, n) = W[k][n] (input gradient, N = in, K = out).
static bool tall_shape(int N, int K) { return N > 128 && N <= 256 && K == 256; }
#ifdef OMDS_TEST_HOOKS
// Test hook (include/omds_test.h, libomds_hip_test.so only): every product of the trainer on the general kernel k_gemm, so that a test
// can hold the special-shape kernels (k_gemm_tall, k_gemm_thin*, k_wgrad_thin) to its bits.  Process-wide.
static std::atomic<int> g_trainer_general{0};
extern "C" OMDS_API int omds_debug_trainer_general_gemm(int on) { g_trainer_general.store(on ? 1 : 0); return OMDS_OK; }
#define OMDS_TRAINER_GENERAL() (g_trainer_general.load() != 0)
#else
#define OMDS_TRAINER_GENERAL() false
#endif
// pack = the caller's 256 KB fragment buffer (one per trainer: a launch re-packs it on the trainer's own stream -- the weights change
// every step -- so two trainers on one device never share it)
template <int EPI>
static int launch_gemm_tall(hipStream_t s, float4* pack, const float* A, int lda, const float* W, int ldw, int trans, float* C, int ldc, int M, int N, int K,
                            const float* aux, int act) {
    static std::atomic<uint64_t> configured{0};    // per device: function attributes belong to the device the kernel is loaded on
    static const int dbg = OMDS_EXP_ENV("OMDS_TALL_DBG", 0);   // experiment builds: 1 no A traffic, 2 no C traffic, 4 no mask traffic, 8 the general kernel instead
    const int lds = 2 * TALL_BUF;
    if ((dbg & 8) || OMDS_TRAINER_GENERAL()) return 1;
    if (!pack || lda != 256 || K != 256 || (reinterpret_cast<size_t>(A) & 15)) return 1;   // the general kernel takes it
    if (omds_first_use_on_device(configured)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tall<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tall<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tall<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    hipLaunchKernelGGL(k_pack256, dim3(8 * 32 * 64 / 256), dim3(256), 0, s, W, ldw, N, K, trans, pack);
    const int ntiles = (M + TALL_ROWS - 1) / TALL_ROWS;
    hipLaunchKernelGGL((k_gemm_tall<EPI>), dim3((unsigned)std::min(ntiles, omds_cu_count())), dim3(512), lds, s, A, pack, C, ldc, M, N, aux, act, ntiles, dbg);
    return 0;
}

static bool thin_off() { static const int dbg = OMDS_EXP_ENV("OMDS_TALL_DBG", 0); return (dbg & 16) != 0 || OMDS_TRAINER_GENERAL(); }   // experiment builds: bit 16 = the general kernel everywhere
template <int EPI>
static bool launch_gemm_thin(hipStream_t s, const float* A, int lda, const float* W, int ldw, int trans, float* C, int ldc, int M, int N, int K,
                             const float* aux, int act) {
    if (K > THIN_KMAX || N > 256 || N < 64 || M < 1 || thin_off()) return false;   // (N < 64: the thin-output products have a kernel of their own)
    const int chunks = (M + THIN_ROWS - 1) / THIN_ROWS;
    hipLaunchKernelGGL((k_gemm_thin<EPI>), dim3((unsigned)std::min(chunks, 8 * omds_cu_count())), dim3(256), 0, s, A, lda, W, ldw, trans, C, ldc, M, N, K, aux, act);
    return true;
}

static int forward(omds_trainer* tr, int B, const float* x) {
    hipStream_t s = tr->stream;
    const int d = tr->d;
    hipLaunchKernelGGL(k_encode, dim3((unsigned)(((size_t)B * d + 255) / 256)), dim3(256), 0, s, x, B, d, tr->H[0]);
    for (int i = 0; i < tr->L; ++i) {
        const int in = tr->dims[i], out = tr->dims[i + 1];
        const int a = i + 1 < tr->L ? tr->act : -1;
        if (launch_gemm_thin<1>(s, tr->H[i], in, tr->W[i], in, 0, tr->H[i + 1], out, B, out, in, tr->b[i], a)) continue;
        if (out <= THINN_NMAX && in <= 256 && in > THIN_KMAX && !thin_off()) {   // the thin-output layer
            static std::atomic<uint64_t> configured{0};
            const int lds = (THINN_ROWS + THINN_NMAX) * LDH * 4;
            if (omds_first_use_on_device(configured)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_thin_out), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            const int chunks = (B + THINN_ROWS - 1) / THINN_ROWS;
            hipLaunchKernelGGL(k_gemm_thin_out, dim3((unsigned)std::min(chunks, 12 * omds_cu_count())), dim3(128), (THINN_ROWS + out) * LDH * 4, s, tr->H[i], in, tr->W[i], in,
                               tr->H[i + 1], out, B, out, in, tr->b[i], a);
            continue;
        }
        if (!tall_shape(out, in) || launch_gemm_tall<1>(s, tr->pack256, tr->H[i], in, tr->W[i], in, 0, tr->H[i + 1], out, B, out, in, tr->b[i], a))
            launch_gemm<false, true, 1>(s, tr->H[i], in, tr->W[i], in, tr->H[i + 1], out, B, out, in, 1, in, 0, tr->b[i], a);
    }
    TCK(hipGetLastError());   // an invalid launch configuration must not go on as a loss computed from stale buffers
    return OMDS_OK;
}

static int mse(omds_trainer* tr, int B, const float* y, float* G, double* loss) {
    const size_t n = (size_t)B * tr->dims[tr->L];
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(k_mse, dim3(blocks), dim3(256), 0, tr->stream, tr->H[tr->L], y, n, 2.f / (float)n, G, tr->lossp);
    TCK(hipGetLastError());
    TCK(hipMemcpyAsync(tr->h_lossp, tr->lossp, (size_t)blocks * 8, hipMemcpyDeviceToHost, tr->stream));
    TCK(hipStreamSynchronize(tr->stream));
    double sum = 0.0;
    for (int i = 0; i < blocks; ++i) sum += tr->h_lossp[i];
    *loss = sum / (double)n;
    return OMDS_OK;
}

extern "C" {

const char* omds_trainer_last_error(const omds_trainer* tr) { return tr ? tr->err.c_str() : g_train_err.c_str(); }

int omds_trainer_create(int device, int n_linear, const int32_t* dims, int act, omds_trainer** out) {
    if (!out || !dims || n_linear < 2 || n_linear > OMDS_MAX_HIDDEN + 1 || (act != OMDS_ACT_RELU && act != OMDS_ACT_TANH) || dims[0] % 3 != 0) {
        g_train_err = "omds_trainer_create: need 2 <= n_linear <= 9 Linear layers, dims[0] = 3 * (raw inputs), act RELU or TANH";
        return OMDS_ERR_INVALID_ARG;
    }
    for (int i = 0; i <= n_linear; ++i)
        if (dims[i] < 1 || dims[i] > 4096) { g_train_err = "omds_trainer_create: layer widths must be in 1 .. 4096"; return OMDS_ERR_INVALID_ARG; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_train_err = std::string("omds_trainer_create: no HIP device available (") + hipGetErrorString(e) + "); this library has no CPU fallback";
        return OMDS_ERR_HIP;
    }
    if (device < 0 || device >= ndev) { g_train_err = "omds_trainer_create: device ordinal out of range"; return OMDS_ERR_INVALID_ARG; }
    omds_trainer* tr = new (std::nothrow) omds_trainer();
    if (!tr) { g_train_err = "omds_trainer_create: out of host memory"; return OMDS_ERR_INVALID_ARG; }
    tr->dev = device; tr->L = n_linear; tr->act = act; tr->d = dims[0] / 3;
    tr->dims.assign(dims, dims + n_linear + 1);
    auto fail = [&](const char* what, hipError_t err) { g_train_err = std::string(what) + ": " + hipGetErrorString(err); omds_trainer_destroy(tr); return OMDS_ERR_HIP; };
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&tr->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    size_t wmax = 0;
    for (int i = 0; i < n_linear; ++i) {
        const size_t nw = (size_t)dims[i] * dims[i + 1], nb = dims[i + 1];
        wmax = std::max(wmax, nw);
        for (auto* vec : {&tr->W, &tr->gW, &tr->mW, &tr->vW}) {
            float* p = nullptr;
            if ((e = hipMalloc(&p, nw * 4)) != hipSuccess) return fail("hipMalloc", e);
            (void)hipMemsetAsync(p, 0, nw * 4, tr->stream);
            vec->push_back(p);
        }
        for (auto* vec : {&tr->b, &tr->gb, &tr->mb, &tr->vb}) {
            float* p = nullptr;
            if ((e = hipMalloc(&p, nb * 4)) != hipSuccess) return fail("hipMalloc", e);
            (void)hipMemsetAsync(p, 0, nb * 4, tr->stream);
            vec->push_back(p);
        }
    }
    tr->H.assign(n_linear + 1, nullptr);
    tr->partial_floats = 256 * wmax;     // up to 256 split-K partials of the largest weight gradient
    if ((e = hipMalloc(&tr->partial, tr->partial_floats * 4)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc(&tr->pack256, 8 * 32 * 64 * sizeof(float4))) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc(&tr->lossp, 1024 * 8)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipHostMalloc(&tr->h_lossp, 1024 * 8)) != hipSuccess) return fail("hipHostMalloc", e);
    (void)hipStreamSynchronize(tr->stream);
    *out = tr;
    return OMDS_OK;
}

void omds_trainer_destroy(omds_trainer* tr) {
    if (!tr) return;
    (void)hipSetDevice(tr->dev);
    if (tr->stream) (void)hipStreamSynchronize(tr->stream);
    trainer_free_data(tr);
    for (auto* vec : {&tr->W, &tr->b, &tr->gW, &tr->gb, &tr->mW, &tr->mb, &tr->vW, &tr->vb})
        for (float* p : *vec) if (p) (void)hipFree(p);
    if (tr->partial) (void)hipFree(tr->partial);
    if (tr->pack256) (void)hipFree(tr->pack256);
    if (tr->lossp) (void)hipFree(tr->lossp);
    if (tr->h_lossp) (void)hipHostFree(tr->h_lossp);
    if (tr->stream) (void)hipStreamDestroy(tr->stream);
    delete tr;
}

int omds_trainer_set_weights(omds_trainer* tr, const float* const* W, const float* const* b) {
    if (!tr) return OMDS_ERR_INVALID_ARG;
    if (!W || !b) { tr->err = "omds_trainer_set_weights: null argument"; return OMDS_ERR_INVALID_ARG; }
    TCK(hipSetDevice(tr->dev));
    for (int i = 0; i < tr->L; ++i) {
        const size_t nw = (size_t)tr->dims[i] * tr->dims[i + 1], nb = tr->dims[i + 1];
        TCK(hipMemcpy(tr->W[i], W[i], nw * 4, hipMemcpyHostToDevice));
        TCK(hipMemcpy(tr->b[i], b[i], nb * 4, hipMemcpyHostToDevice));
        for (float* p : {tr->mW[i], tr->vW[i]}) TCK(hipMemset(p, 0, nw * 4));   // a fresh optimizer state, like a new torch.optim.Adam
        for (float* p : {tr->mb[i], tr->vb[i]}) TCK(hipMemset(p, 0, nb * 4));
    }
    tr->step = 0;
    return OMDS_OK;
}

int omds_trainer_get_weights(omds_trainer* tr, float* const* W, float* const* b) {
    if (!tr) return OMDS_ERR_INVALID_ARG;
    if (!W || !b) { tr->err = "omds_trainer_get_weights: null argument"; return OMDS_ERR_INVALID_ARG; }
    TCK(hipSetDevice(tr->dev));
    TCK(hipStreamSynchronize(tr->stream));
    for (int i = 0; i < tr->L; ++i) {
        TCK(hipMemcpy(W[i], tr->W[i], (size_t)tr->dims[i] * tr->dims[i + 1] * 4, hipMemcpyDeviceToHost));
        TCK(hipMemcpy(b[i], tr->b[i], (size_t)tr->dims[i + 1] * 4, hipMemcpyDeviceToHost));
    }
    return OMDS_OK;
}

int omds_trainer_set_data(omds_trainer* tr, const float* x, const float* y, int batch) {
    if (!tr) return OMDS_ERR_INVALID_ARG;
    if (!x || !y || batch < 1) { tr->err = "omds_trainer_set_data: need batch >= 1 and non-null x [B, d], y [B, C]"; return OMDS_ERR_INVALID_ARG; }
    TCK(hipSetDevice(tr->dev));
    TCK(hipStreamSynchronize(tr->stream));
    if (batch > tr->xcap) {
        for (float* p : {tr->x, tr->y}) if (p) (void)hipFree(p);
        tr->x = tr->y = nullptr; tr->xcap = 0; tr->B = 0;
        TCK(hipMalloc(&tr->x, (size_t)batch * tr->d * 4));
        TCK(hipMalloc(&tr->y, (size_t)batch * tr->dims[tr->L] * 4));
        tr->xcap = batch;
    }
    int rc;
    if ((rc = trainer_reserve(tr, batch))) return rc;
    TCK(hipMemcpy(tr->x, x, (size_t)batch * tr->d * 4, hipMemcpyHostToDevice));
    TCK(hipMemcpy(tr->y, y, (size_t)batch * tr->dims[tr->L] * 4, hipMemcpyHostToDevice));
    tr->B = batch;
    return OMDS_OK;
}

// The validation split (train_sdf.py:84-86, 117-121) beside the training set: evaluated by omds_trainer_eval(which = 1) with the
// trainer's current weights -- no weight copy, no second trainer, the optimizer state untouched.
int omds_trainer_set_val_data(omds_trainer* tr, const float* x, const float* y, int batch) {
    if (!tr) return OMDS_ERR_INVALID_ARG;
    if (!x || !y || batch < 1) { tr->err = "omds_trainer_set_val_data: need batch >= 1 and non-null x [B, d], y [B, C]"; return OMDS_ERR_INVALID_ARG; }
    TCK(hipSetDevice(tr->dev));
    TCK(hipStreamSynchronize(tr->stream));
    if (batch > tr->vcap) {
        for (float* p : {tr->xv, tr->yv}) if (p) (void)hipFree(p);
        tr->xv = tr->yv = nullptr; tr->vcap = 0; tr->Bv = 0;
        TCK(hipMalloc(&tr->xv, (size_t)batch * tr->d * 4));
        TCK(hipMalloc(&tr->yv, (size_t)batch * tr->dims[tr->L] * 4));
        tr->vcap = batch;
    }
    int rc;
    if ((rc = trainer_reserve(tr, std::max(batch, tr->B)))) return rc;
    TCK(hipMemcpy(tr->xv, x, (size_t)batch * tr->d * 4, hipMemcpyHostToDevice));
    TCK(hipMemcpy(tr->yv, y, (size_t)batch * tr->dims[tr->L] * 4, hipMemcpyHostToDevice));
    tr->Bv = batch;
    return OMDS_OK;
}

// forward + F.mse_loss with the current weights, no update: which = 0 the training set, 1 the validation set (train_sdf.py:117-121)
int omds_trainer_eval(omds_trainer* tr, int which, float* mse_out, float* pred_out) {
    if (!tr) return OMDS_ERR_INVALID_ARG;
    if (which != 0 && which != 1) { tr->err = "omds_trainer_eval: which must be 0 (training set) or 1 (validation set)"; return OMDS_ERR_INVALID_ARG; }
    const int B = which ? tr->Bv : tr->B;
    if (B < 1) { tr->err = which ? "omds_trainer_eval: no validation set (omds_trainer_set_val_data)" : "omds_trainer_eval: no data set (omds_trainer_set_data)"; return OMDS_ERR_NOT_INITIALISED; }
    TCK(hipSetDevice(tr->dev));
    int rc;
    if ((rc = trainer_reserve(tr, B))) return rc;
    if ((rc = forward(tr, B, which ? tr->xv : tr->x))) return rc;
    double loss = 0.0;
    if ((rc = mse(tr, B, which ? tr->yv : tr->y, nullptr, &loss))) return rc;
    if (mse_out) *mse_out = (float)loss;
    if (pred_out) TCK(hipMemcpy(pred_out, tr->H[tr->L], (size_t)B * tr->dims[tr->L] * 4, hipMemcpyDeviceToHost));
    return OMDS_OK;
}

// torch.optim.Adam's state (exp_avg, exp_avg_sq per parameter, the step count), for checkpoints in the reference's format
// (train_sdf.py:130-138 saves optimizer.state_dict()) and for resuming: arrays like omds_trainer_get_weights (NULL array = skip).
int omds_trainer_get_optimizer_state(omds_trainer* tr, float* const* mW, float* const* mb, float* const* vW, float* const* vb, int64_t* step) {
    if (!tr) return OMDS_ERR_INVALID_ARG;
    TCK(hipSetDevice(tr->dev));
    TCK(hipStreamSynchronize(tr->stream));
    for (int i = 0; i < tr->L; ++i) {
        const size_t nw = (size_t)tr->dims[i] * tr->dims[i + 1] * 4, nb = (size_t)tr->dims[i + 1] * 4;
        if (mW && mW[i]) TCK(hipMemcpy(mW[i], tr->mW[i], nw, hipMemcpyDeviceToHost));
        if (vW && vW[i]) TCK(hipMemcpy(vW[i], tr->vW[i], nw, hipMemcpyDeviceToHost));
        if (mb && mb[i]) TCK(hipMemcpy(mb[i], tr->mb[i], nb, hipMemcpyDeviceToHost));
        if (vb && vb[i]) TCK(hipMemcpy(vb[i], tr->vb[i], nb, hipMemcpyDeviceToHost));
    }
    if (step) *step = tr->step;
    return OMDS_OK;
}
int omds_trainer_set_optimizer_state(omds_trainer* tr, const float* const* mW, const float* const* mb, const float* const* vW,
                                     const float* const* vb, int64_t step) {
    if (!tr) return OMDS_ERR_INVALID_ARG;
    if (!mW || !mb || !vW || !vb || step < 0) { tr->err = "omds_trainer_set_optimizer_state: null argument or negative step"; return OMDS_ERR_INVALID_ARG; }
    for (int i = 0; i < tr->L; ++i)
        if (!mW[i] || !mb[i] || !vW[i] || !vb[i]) { tr->err = "omds_trainer_set_optimizer_state: null array"; return OMDS_ERR_INVALID_ARG; }
    TCK(hipSetDevice(tr->dev));
    TCK(hipStreamSynchronize(tr->stream));
    for (int i = 0; i < tr->L; ++i) {
        const size_t nw = (size_t)tr->dims[i] * tr->dims[i + 1] * 4, nb = (size_t)tr->dims[i + 1] * 4;
        TCK(hipMemcpy(tr->mW[i], mW[i], nw, hipMemcpyHostToDevice));
        TCK(hipMemcpy(tr->vW[i], vW[i], nw, hipMemcpyHostToDevice));
        TCK(hipMemcpy(tr->mb[i], mb[i], nb, hipMemcpyHostToDevice));
        TCK(hipMemcpy(tr->vb[i], vb[i], nb, hipMemcpyHostToDevice));
    }
    tr->step = step;
    return OMDS_OK;
}

// one epoch of train_sdf.py:105-113 on the whole data set: forward, mse_loss, backward, Adam step.  loss_out = the loss BEFORE the step
int omds_trainer_step(omds_trainer* tr, float lr, float beta1, float beta2, float eps, float* loss_out) {
    if (!tr) return OMDS_ERR_INVALID_ARG;
    if (tr->B < 1) { tr->err = "omds_trainer_step: no data set (omds_trainer_set_data)"; return OMDS_ERR_NOT_INITIALISED; }
    TCK(hipSetDevice(tr->dev));
    hipStream_t s = tr->stream;
    const int B = tr->B, L = tr->L;
    int rc;
    if ((rc = trainer_reserve(tr, B))) return rc;
    if ((rc = forward(tr, B, tr->x))) return rc;
    double loss = 0.0;
    float* G = tr->G[0];
    float* Gn = tr->G[1];
    if ((rc = mse(tr, B, tr->y, G, &loss))) return rc;
    if (loss_out) *loss_out = (float)loss;
    for (int i = L - 1; i >= 0; --i) {
        const int in = tr->dims[i], out = tr->dims[i + 1];
        // dW = G^T . H_i: the contraction runs over the batch; split it so that ~1024 workgroups exist, sum the partials in order
        const int tiles = ((out + TM - 1) / TM) * ((in + TN - 1) / TN);
        int splits = std::max(1, std::min({256, 1024 / std::max(tiles, 1), (B + 4 * TK - 1) / (4 * TK)}));
        while ((size_t)splits * out * in > tr->partial_floats) --splits;
        int kchunk = ((B + splits - 1) / splits + TK - 1) / TK * TK;
        splits = (B + kchunk - 1) / kchunk;
        // (the thin kernels keep the general kernel's chunk boundaries: the same partial sums, summed in the same order)
        if (in <= WG_TMAX && out <= 4096 && out > WG_TMAX && !thin_off())        // first layer: thin = H_i [B x in] (dW's columns), wide = G [B x out]
            launch_wgrad_thin(s, (out + 255) / 256, splits, G, out, out, tr->H[i], in, in, 1, tr->partial, B, kchunk, (size_t)out * in, in);
        else if (out <= WG_TMAX && in <= 4096 && in > WG_TMAX && !thin_off())    // last layer: thin = G [B x out] (dW's rows), wide = H_i [B x in]
            launch_wgrad_thin(s, (in + 255) / 256, splits, tr->H[i], in, in, G, out, out, 0, tr->partial, B, kchunk, (size_t)out * in, in);
        else
            launch_gemm<true, false, 0>(s, G, out, tr->H[i], in, tr->partial, in, out, in, B, splits, kchunk, (size_t)out * in);
        hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)(((size_t)out * in + 255) / 256)), dim3(256), 0, s, tr->partial, splits, (size_t)out * in, tr->gW[i]);
        // db = column sums of G
        const int rsplit = std::max(1, std::min({2048, B / 256, (int)(tr->partial_floats / (size_t)out)}));
        const size_t rows_per = ((size_t)B + rsplit - 1) / rsplit;
        hipLaunchKernelGGL(k_colsum_partial, dim3((out + 255) / 256, rsplit), dim3(256), 0, s, G, (size_t)B, out, rows_per, tr->partial);
        hipLaunchKernelGGL(k_sum_partials, dim3((out + 255) / 256), dim3(256), 0, s, tr->partial, rsplit, (size_t)out, tr->gb[i]);
        if (i > 0) {   // gradient at this layer's input, through the activation of the layer in front
            if (launch_gemm_thin<2>(s, G, out, tr->W[i], in, 1, Gn, in, B, in, out, tr->H[i], tr->act)) { std::swap(G, Gn); continue; }
            if (!tall_shape(in, out) || launch_gemm_tall<2>(s, tr->pack256, G, out, tr->W[i], in, 1, Gn, in, B, in, out, tr->H[i], tr->act))
                launch_gemm<false, false, 2>(s, G, out, tr->W[i], in, Gn, in, B, in, out, 1, out, 0, tr->H[i], tr->act);
            std::swap(G, Gn);
        }
    }
    tr->step++;
    const double bc1 = 1.0 - std::pow((double)beta1, (double)tr->step), bc2 = 1.0 - std::pow((double)beta2, (double)tr->step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)std::sqrt(bc2);
    for (int i = 0; i < L; ++i) {
        const size_t nw = (size_t)tr->dims[i] * tr->dims[i + 1], nb = tr->dims[i + 1];
        hipLaunchKernelGGL(k_adam, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, tr->W[i], tr->gW[i], tr->mW[i], tr->vW[i], nw, beta1, beta2, eps, step_size, bc2_sqrt);
        hipLaunchKernelGGL(k_adam, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, tr->b[i], tr->gb[i], tr->mb[i], tr->vb[i], nb, beta1, beta2, eps, step_size, bc2_sqrt);
    }
    TCK(hipGetLastError());
    TCK(hipStreamSynchronize(s));
    return OMDS_OK;
}

}  // extern "C"

// The GEMM for callers outside the trainer (wide_kernels.hip: distance networks wider than the fused kernels' 256 columns)
void omds_launch_linear_forward(hipStream_t s, const float* H, int in, const float* W, const float* b, float* Out, int out, int B, int act) {
    launch_gemm<false, true, 1>(s, H, in, W, in, Out, out, B, out, in, 1, in, 0, b, act);
}
void omds_launch_linear_inputgrad(hipStream_t s, const float* G, int out, const float* W, int in, float* Gi, int B, const float* Hact, int act) {
    if (Hact) launch_gemm<false, false, 2>(s, G, out, W, in, Gi, in, B, in, out, 1, out, 0, Hact, act);
    else launch_gemm<false, false, 0>(s, G, out, W, in, Gi, in, B, in, out, 1, out, 0);
}

// Distance-network kernels for gfx950 (MI355X): exact-fp32 MFMA, activations resident in LDS.
//
// What they replace in the reference (paths relative to python_scripts/):
//   k_pass1          : MPPI.build_nn_input + model_jit.forward over all N*O (rollout, obstacle)
//                      pairs + /100, -radius, link mask, min over links
//                      (ds_mppi/functions/MPPI.py:93-95, 233-243; mlp_learn/sdf/network_macros_mod.py:137-146)
//   k_topk           : mindist_matrix.sort(dim=1)[:, :k]                       (MPPI.py:245-247)
//   k_pass2          : RobotSdfCollisionNet.functorch_vjp on the N*k closest rows: forward,
//                      arg-min link over all raw outputs, analytic backward (ReLU masks + the
//                      positional-encoding chain rule)                         (robot_sdf.py:153-158)
//   k_blend          : softmax(-10 d) blend of the k gradients                 (MPPI.py:270-280)
//
// Design (MI355X-first, not a translation):
//   * layer 1 is separable: z1[t,o] = (W1_q f(q_t) + b1) + W1_p f(p_o) = Apre[t] + Bpre[o], so the
//     [N*O, n+4] input matrix is never materialised and sin/cos are evaluated per rollout and
//     per obstacle, not per pair;
//   * a workgroup owns MT consecutive rows of the virtual (rollout-major) row space, keeps the
//     [MT x 256] activation tile in LDS (row stride 260 floats: conflict-free ds_read_b128 of
//     MFMA A-fragments) across all layers, and streams the pre-packed weights straight from L2
//     into VGPRs as MFMA B-fragments (1 KiB fully coalesced per wave-load, no LDS round trip);
//   * v_mfma_f32_32x32x2_f32: the K order inside a dot product is free, so each lane loads four
//     consecutive k (one ds_read_b128 / global_load_dwordx4) and feeds four MFMA steps;
//   * accumulators stay in registers until the whole layer is done, so the tile is updated in
//     place (one LDS buffer, two barriers per layer).
#include "mlp_device.h"

// ------------------------------------------------------------------------------------------------
// layer-1 halves
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rollout_layer1(MlpDev m, const float* __restrict__ qT, int ldq, int B,
                                                        float* __restrict__ Apre) {
    __shared__ float f[3 * OMDS_MAX_DOF];
    const int t = blockIdx.x, c = threadIdx.x, n = m.n_dof, d = m.d;
    if (c < n) {
        const float q = qT[(size_t)c * ldq + t];
        f[c] = q;
        f[n + c] = sinf(q);
        f[2 * n + c] = cosf(q);
    }
    __syncthreads();
    float acc = m.b1[c];
    for (int j = 0; j < n; ++j) acc = fmaf(m.W1t[(size_t)j * OMDS_WIDTH + c], f[j], acc);
    for (int j = 0; j < n; ++j) acc = fmaf(m.W1t[(size_t)(d + j) * OMDS_WIDTH + c], f[n + j], acc);
    for (int j = 0; j < n; ++j) acc = fmaf(m.W1t[(size_t)(2 * d + j) * OMDS_WIDTH + c], f[2 * n + j], acc);
    Apre[(size_t)t * OMDS_WIDTH + c] = acc;
}

__global__ __launch_bounds__(256) void k_obstacle_layer1(MlpDev m, const float* __restrict__ xyzr, int O,
                                                         float* __restrict__ Bpre, float* __restrict__ radius) {
    __shared__ float f[9];
    const int o = blockIdx.x, c = threadIdx.x, n = m.n_dof, d = m.d;
    if (c < 3) {
        const float p = xyzr[o * 4 + c];
        f[c] = p;
        f[3 + c] = sinf(p);
        f[6 + c] = cosf(p);
    }
    if (c == 3) radius[o] = xyzr[o * 4 + 3];
    __syncthreads();
    float acc = 0.f;
    for (int j = 0; j < 3; ++j) acc = fmaf(m.W1t[(size_t)(n + j) * OMDS_WIDTH + c], f[j], acc);
    for (int j = 0; j < 3; ++j) acc = fmaf(m.W1t[(size_t)(d + n + j) * OMDS_WIDTH + c], f[3 + j], acc);
    for (int j = 0; j < 3; ++j) acc = fmaf(m.W1t[(size_t)(2 * d + n + j) * OMDS_WIDTH + c], f[6 + j], acc);
    Bpre[(size_t)o * OMDS_WIDTH + c] = acc;
}

template <int MT, int MR, int NR>
struct Geo {
    static constexpr int WM = MT / (32 * MR);
    static constexpr int WN = OMDS_NCB / NR;
    static constexpr int NW = WM * WN;
    static constexpr int NT = NW * 64;
    static_assert(WM >= 1 && WN >= 1 && WM * 32 * MR == MT && WN * NR == OMDS_NCB, "bad tile geometry");
};

// ------------------------------------------------------------------------------------------------
// pass 1: all (rollout, obstacle) pairs -> min link distance
// ------------------------------------------------------------------------------------------------
template <int MT, int MR, int NR, int ACT>
__device__ __forceinline__ void pass1_tile(const MlpDev& m, float* smem, const float* __restrict__ Apre,
                                           const float* __restrict__ Bpre, const float* __restrict__ radius, int O,
                                           long long total_rows, uint32_t ignored, float* __restrict__ Dmin, int tune,
                                           const long long row0) {
    using G = Geo<MT, MR, NR>;
    float* Hs = smem;                                           // [MT][LDH]
    float* rowRad = smem + MT * LDH;                            // [MT] obstacle radius of each row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / G::WN, wn = wave % G::WN;

    // ---- layer 1: H1 = act(Apre[t] + Bpre[o]); one float4 per thread and iteration, all loads of the
    //      tile issued before the first use (the loop is fully unrolled, no 64-bit division per row) -------
    {
        constexpr int IT = MT * 64 / G::NT;                     // iterations per thread
        const long long t0 = row0 / O;                          // wave-uniform, once per workgroup
        const int o0 = (int)(row0 - t0 * O);
        const int rows_here = (int)((total_rows - row0 < MT) ? (total_rows - row0) : MT);
        constexpr int BI = IT < 8 ? IT : 8;                     // loads in flight per thread and batch
#pragma unroll 1
        for (int base = 0; base < IT; base += BI) {
            float4 av[BI], bv[BI];
            int oo[BI];
#pragma unroll
            for (int it = 0; it < BI; ++it) {
                const int idx = tid + (base + it) * G::NT;
                const int r = idx >> 6, c4 = idx & 63;
                const int oq = o0 + r;                          // < O + MT
                const int dt = oq / O;                          // 32-bit
                const int o = oq - dt * O;
                oo[it] = o;
                if (r < rows_here) {
                    av[it] = reinterpret_cast<const float4*>(Apre)[(t0 + dt) * 64 + c4];
                    bv[it] = reinterpret_cast<const float4*>(Bpre)[(size_t)o * 64 + c4];
                } else {
                    av[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                    bv[it] = av[it];
                }
            }
#pragma unroll
            for (int it = 0; it < BI; ++it) {
                const int idx = tid + (base + it) * G::NT;
                const int r = idx >> 6, c4 = idx & 63;
                float4 v;
                v.x = actf(av[it].x + bv[it].x, ACT);
                v.y = actf(av[it].y + bv[it].y, ACT);
                v.z = actf(av[it].z + bv[it].z, ACT);
                v.w = actf(av[it].w + bv[it].w, ACT);
                if (r >= rows_here) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(Hs + r * LDH + 4 * c4) = v;
                if (c4 == 0) rowRad[r] = (r < rows_here) ? radius[oo[it]] : 0.f;
            }
        }
    }
    __syncthreads();

    // ---- hidden -> hidden layers -----------------------------------------------------------------
    const float* Hw = Hs + (wm * MR * 32) * LDH;
    const int cb0 = wn * NR;
    for (int l = 0; l < m.nhh; ++l) {
        f32x16 acc[MR][NR];
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        float bvj[NR];   // biases fetched before the GEMM so that the epilogue does not start with an L2 round trip
#pragma unroll
        for (int j = 0; j < NR; ++j) bvj[j] = m.bh[l * OMDS_WIDTH + (cb0 + j) * 32 + (lane & 31)];
        gemm256<MR, NR>(Hw, m.Wf + (size_t)l * (OMDS_NCB * 32 * 64), cb0, lane, acc, (tune & 1) != 0);
        __syncthreads();  // every wave has finished reading the tile
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = (cb0 + j) * 32 + (lane & 31);
            const float bv = bvj[j];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Hs[((wm * MR + i) * 32 + crow(r, lane)) * LDH + col] = actf(acc[i][j][r] + bv, ACT);
        }
        __syncthreads();
    }

    // ---- last layer (256 -> C, padded to 16) on v_mfma_f32_16x16x4_f32, 16 rows per wave ----------
    for (int rb = wave; rb < MT / 16; rb += G::NW) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* arow = Hs + (rb * 16 + (lane & 15)) * LDH + 4 * (lane >> 4);
#pragma unroll 4
        for (int c = 0; c < 16; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(arow + 16 * c);
            const float4 w = m.Wl[c * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
        }
        // C/D layout 16x16: col = lane&15 (link), row = 4(lane>>4) + reg
        const int j = lane & 15;
        const float bj = m.bl[j];
        const bool pad = j >= m.C, ign = (ignored >> j) & 1u;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = rb * 16 + 4 * (lane >> 4) + reg;
            float y = (acc[reg] + bj) / m.out_div - rowRad[r];
            y = pad ? __builtin_inff() : (ign ? 1e6f : y);
            y = fminf(y, __shfl_xor(y, 1));
            y = fminf(y, __shfl_xor(y, 2));
            y = fminf(y, __shfl_xor(y, 4));
            y = fminf(y, __shfl_xor(y, 8));
            if (j == 0 && row0 + r < total_rows) Dmin[row0 + r] = y;
        }
    }
}

template <int MT, int MR, int NR, int ACT>
__global__ __launch_bounds__((Geo<MT, MR, NR>::NT)) void k_pass1(MlpDev m, const float* __restrict__ Apre,
                                                               const float* __restrict__ Bpre,
                                                               const float* __restrict__ radius, int O,
                                                               long long total_rows, uint32_t ignored,
                                                               float* __restrict__ Dmin, int tune) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    pass1_tile<MT, MR, NR, ACT>(m, smem, Apre, Bpre, radius, O, total_rows, ignored, Dmin, tune, (long long)blockIdx.x * MT);
}

// Mixed-granularity launch: the first n_big workgroups take 64-row tiles, the rest cover the remaining rows
// in 32-row tiles.  Workgroups are dispatched in index order, so the kernel ends on small tiles and the
// drain phase (slots idling while the last tiles finish) shrinks with the tile size.
template <int ACT>
__global__ __launch_bounds__(512) void k_pass1_mixed(MlpDev m, const float* __restrict__ Apre,
                                                     const float* __restrict__ Bpre, const float* __restrict__ radius,
                                                     int O, long long total_rows, uint32_t ignored,
                                                     float* __restrict__ Dmin, int tune, int n_big) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < n_big) {
        pass1_tile<64, 2, 1, ACT>(m, smem, Apre, Bpre, radius, O, total_rows, ignored, Dmin, tune, (long long)b * 64);
    } else {
        pass1_tile<32, 1, 1, ACT>(m, smem, Apre, Bpre, radius, O, total_rows, ignored, Dmin, tune,
                                  (long long)n_big * 64 + (long long)(b - n_big) * 32);
    }
}

// ------------------------------------------------------------------------------------------------
// top-k (ascending, ties by lower obstacle index): one wave per rollout
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_topk(const float* __restrict__ Dmin, int B, int O, int k,
                                              int32_t* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= B) return;
    topk_row(Dmin + (size_t)t * O, O, k, lane, [&](int j, int bi) { idx[(size_t)t * k + j] = bi; });
}

// ------------------------------------------------------------------------------------------------
// pass 2: forward + analytic backward on the N*k closest rows. 32 rows per workgroup, 8 waves,
// wave w owns output columns [32w, 32w+32). ReLU masks live in LDS as 16 bits per thread/layer
// in the MFMA C-layout (the same lane owns the same (row, col) in every layer).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(P2_NT) void k_pass2(MlpDev m, const float* __restrict__ Apre,
                                                 const float* __restrict__ Bpre, const float* __restrict__ radius,
                                                 const float* __restrict__ xyzr, const int32_t* __restrict__ idx,
                                                 int total_rows, int k, const float* __restrict__ qT, int ldq,
                                                 float* __restrict__ gradx, float* __restrict__ drow,
                                                 float* __restrict__ yraw, int32_t* __restrict__ minidx,
                                                 float* __restrict__ dscr) {
    // dscr (tanh only): [nhh+1][rows padded to 32][256] activation derivatives 1 - h^2, L2-resident scratch
    extern __shared__ __attribute__((aligned(16))) float smem[];
    P2Smem sm;
    sm.Hs = smem;
    sm.P = sm.Hs + P2_MT * LDH;
    sm.gf = sm.P + 8 * 32 * 33;
    sm.maskL = reinterpret_cast<uint32_t*>(sm.gf + 32 * 33);
    sm.rowT = reinterpret_cast<int*>(sm.maskL + (m.nhh + 1) * P2_NT);
    sm.rowO = sm.rowT + P2_MT;
    sm.rowMin = sm.rowO + P2_MT;
    const int tid = threadIdx.x;
    const int R0 = blockIdx.x * P2_MT;
    if (tid < P2_MT) {
        const int R = R0 + tid;
        int t = -1, o = 0;
        if (R < total_rows) { t = R / k; o = idx[R]; }
        sm.rowT[tid] = t;
        sm.rowO[tid] = o;
    }
    __syncthreads();
    pass2_body(m, sm, Apre, Bpre, radius, xyzr, R0, total_rows, qT, ldq, gradx, drow, R0, yraw, minidx, dscr,
               (size_t)gridDim.x * P2_MT * OMDS_WIDTH, R0);
}

// ------------------------------------------------------------------------------------------------
// blend of the k closest gradients (MPPI.py:270-280), standalone form for omds_dist_grad
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blend(const float* __restrict__ gradx, const float* __restrict__ drow, int B,
                                               int k, int d, int n, float softmax_k, float* __restrict__ dist,
                                               float* __restrict__ nngrad) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    float mx = -__builtin_inff();
    for (int j = 0; j < k; ++j) mx = fmaxf(mx, softmax_k * drow[t * k + j]);
    float s = 0.f;
    for (int j = 0; j < k; ++j) s += expf(softmax_k * drow[t * k + j] - mx);
    for (int c = 0; c < n; ++c) {
        float g = 0.f;
        for (int j = 0; j < k; ++j) g += gradx[(size_t)(t * k + j) * d + c] * (expf(softmax_k * drow[t * k + j] - mx) / s);
        nngrad[(size_t)t * n + c] = g;
    }
    dist[t] = drow[t * k];
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void omds_launch_rollout_layer1(hipStream_t s, const MlpDev& m, const float* qT, int ldq, int B, float* Apre) {
    if (B <= 0) return;
    hipLaunchKernelGGL(k_rollout_layer1, dim3(B), dim3(256), 0, s, m, qT, ldq, B, Apre);
}

void omds_launch_obstacle_layer1(hipStream_t s, const MlpDev& m, const float* xyzr, int O, float* Bpre, float* radius) {
    if (O <= 0) return;
    hipLaunchKernelGGL(k_obstacle_layer1, dim3(O), dim3(256), 0, s, m, xyzr, O, Bpre, radius);
}

template <int MT, int MR, int NR, int ACT>
static void launch_pass1_a(hipStream_t s, const MlpDev& m, const float* Apre, const float* Bpre, const float* radius,
                           int O, long long total, uint32_t ignored, float* Dmin) {
    using G = Geo<MT, MR, NR>;
    const size_t lds = (size_t)MT * LDH * 4 + (size_t)MT * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass1<MT, MR, NR, ACT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const long long tiles = (total + MT - 1) / MT;
    static int tune = -1;
    if (tune < 0) { const char* e = getenv("OMDS_TUNE"); tune = e ? atoi(e) : 1; }  // bit 0: s_setprio(1) around MFMA clusters (+1 % measured)
    hipLaunchKernelGGL((k_pass1<MT, MR, NR, ACT>), dim3((unsigned)tiles), dim3(G::NT), lds, s, m, Apre, Bpre, radius, O,
                       total, ignored, Dmin, tune);
}

template <int MT, int MR, int NR>
static void launch_pass1_t(hipStream_t s, const MlpDev& m, const float* Apre, const float* Bpre, const float* radius,
                           int O, long long total, uint32_t ignored, float* Dmin) {
    if (m.act == OMDS_ACT_RELU) launch_pass1_a<MT, MR, NR, OMDS_ACT_RELU>(s, m, Apre, Bpre, radius, O, total, ignored, Dmin);
    else launch_pass1_a<MT, MR, NR, OMDS_ACT_TANH>(s, m, Apre, Bpre, radius, O, total, ignored, Dmin);
}

// Tile choice: 64-row tiles (8 waves, each 64 rows x 32 columns, two workgroups per CU) once there are
// enough tiles to fill 256 CUs; 32-row tiles for small batches (planar configs, dist_grad calls).
template <int ACT>
static void launch_pass1_mixed(hipStream_t s, const MlpDev& m, const float* Apre, const float* Bpre, const float* radius,
                               int O, long long total, uint32_t ignored, float* Dmin, int small_rounds) {
    const size_t lds = (size_t)64 * LDH * 4 + 64 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass1_mixed<ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    static int tune = -1;
    if (tune < 0) { const char* e = getenv("OMDS_TUNE"); tune = e ? atoi(e) : 1; }
    // keep `small_rounds` x 512 x 64 rows (in units of resident 64-row workgroups) for the 32-row tail tiles
    long long tiles64 = total / 64;
    long long keep = (long long)small_rounds * 512 / 2;   // 64-row tiles' worth of rows given to small tiles
    long long n_big = tiles64 > keep ? tiles64 - keep : 0;
    const long long rest = total - n_big * 64;
    const long long n_small = (rest + 31) / 32;
    hipLaunchKernelGGL((k_pass1_mixed<ACT>), dim3((unsigned)(n_big + n_small)), dim3(512), lds, s, m, Apre, Bpre, radius, O,
                       total, ignored, Dmin, tune, (int)n_big);
}

static int g_pass1_variant = -1;  // -1 = auto; set through OMDS_PASS1_VARIANT for experiments
void omds_launch_pass1(hipStream_t s, const MlpDev& m, const float* Apre, const float* Bpre, const float* radius,
                       int O, int B, uint32_t ignored, float* Dmin) {
    const long long total = (long long)B * O;
    if (total <= 0) return;
    if (g_pass1_variant == -1) {
        const char* e = getenv("OMDS_PASS1_VARIANT");
        g_pass1_variant = e ? atoi(e) : 0;
    }
    int v = g_pass1_variant;
    // measured on MI355X (profiles/): 64-row tiles (2 workgroups per CU) beat 128-row tiles at N*O = 301k rows
    // (129 vs 123 TFLOP/s: shorter tail) and tie at 1.2M rows (134 TFLOP/s)
    // and ending the launch on one "round" of 32-row tiles (variant 11) shortens the drain: 135 vs 133 TFLOP/s
    if (v == 0) v = (total >= 64LL * 1024) ? 11 : ((total >= 64LL * 512) ? 3 : 5);
    if (v >= 10) {   // 10 + r: mixed tiles, the last r "rounds" of 512 workgroups use 32-row tiles
        if (m.act == OMDS_ACT_RELU) launch_pass1_mixed<OMDS_ACT_RELU>(s, m, Apre, Bpre, radius, O, total, ignored, Dmin, v - 10);
        else launch_pass1_mixed<OMDS_ACT_TANH>(s, m, Apre, Bpre, radius, O, total, ignored, Dmin, v - 10);
        return;
    }
    switch (v) {
        case 1: launch_pass1_t<128, 4, 1>(s, m, Apre, Bpre, radius, O, total, ignored, Dmin); break;
        case 2: launch_pass1_t<128, 2, 2>(s, m, Apre, Bpre, radius, O, total, ignored, Dmin); break;
        case 3: launch_pass1_t<64, 2, 1>(s, m, Apre, Bpre, radius, O, total, ignored, Dmin); break;
        case 4: launch_pass1_t<64, 2, 2>(s, m, Apre, Bpre, radius, O, total, ignored, Dmin); break;
        case 6: launch_pass1_t<128, 4, 2>(s, m, Apre, Bpre, radius, O, total, ignored, Dmin); break;
        default: launch_pass1_t<32, 1, 1>(s, m, Apre, Bpre, radius, O, total, ignored, Dmin); break;
    }
}

void omds_launch_topk(hipStream_t s, const float* Dmin, int B, int O, int k, int32_t* idx) {
    if (B <= 0) return;
    hipLaunchKernelGGL(k_topk, dim3((B + 3) / 4), dim3(256), 0, s, Dmin, B, O, k, idx);
}

void omds_launch_pass2(hipStream_t s, const MlpDev& m, const float* Apre, const float* Bpre, const float* radius,
                       const float* xyzr, const int32_t* idx, int B, int k, const float* qT, int ldq, float* gradx,
                       float* drow, float* yraw, int32_t* minidx, float* dscr) {
    const int total = B * k;
    if (total <= 0) return;
    const size_t lds = ((size_t)P2_MT * LDH + 8 * 32 * 33 + 32 * 33) * 4 + (size_t)(m.nhh + 1) * P2_NT * 4 + 3 * P2_MT * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(((size_t)P2_MT * LDH + 8 * 32 * 33 + 32 * 33) * 4 +
                                        (size_t)(OMDS_MAX_HIDDEN + 1) * P2_NT * 4 + 3 * P2_MT * 4));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_pass2, dim3((total + P2_MT - 1) / P2_MT), dim3(P2_NT), lds, s, m, Apre, Bpre, radius, xyzr,
                       idx, total, k, qT, ldq, gradx, drow, yraw, minidx, dscr);
}

void omds_launch_blend(hipStream_t s, const float* gradx, const float* drow, int B, int k, int d, int n,
                       float softmax_k, float* dist, float* nngrad) {
    if (B <= 0) return;
    hipLaunchKernelGGL(k_blend, dim3((B + 255) / 256), dim3(256), 0, s, gradx, drow, B, k, d, n, softmax_k, dist, nngrad);
}

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
//   * the [N*O, n+4] input matrix is never materialised: sin/cos are evaluated per rollout and per
//     obstacle (tables Fq / Fp of encoded inputs), a tile gathers its pairs' 30 inputs into LDS and
//     layer 1 is a K = 32 product like every other layer -- the reference's single chain over
//     [q, p, sin q, sin p, cos q, cos p] does not split into a rollout half plus an obstacle half;
//   * a workgroup owns MT consecutive rows of the virtual (rollout-major) row space, keeps the
//     [MT x 256] activation tile in LDS (row stride 260 floats: conflict-free ds_read_b128 of
//     MFMA A-fragments) across all layers, and streams the pre-packed weights straight from L2
//     into VGPRs as MFMA B-fragments (1 KiB fully coalesced per wave-load, no LDS round trip);
//   * v_mfma_f32_32x32x2_f32: each lane loads four consecutive positions (one ds_read_b128 /
//     buffer_load_dwordx4) and feeds four MFMA steps; the tile is stored k-permuted (omds_kpos) so
//     that the resulting chain runs over k in ASCENDING order from zero, bias added last -- the
//     arithmetic of torch-CPU's addmm, bit for bit;
//   * accumulators stay in registers until the whole layer is done, so the tile is updated in
//     place (one LDS buffer, two barriers per layer).
#include "mlp_device.h"

// ------------------------------------------------------------------------------------------------
// encoded inputs [x, sin x, cos x] of the rollout states and of the obstacle points (network_macros_mod.py:139-140), each at its
// feature slot part * d + j of a 32-float row; a pair's input row is the OR of its two rows (pass1_tile, pass2_body)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rollout_features(MlpDev m, const float* __restrict__ qT, int ldq, int B,
                                                          float* __restrict__ Fq, _Float16* __restrict__ FqH, int ldF, int slab) {
    const int t = blockIdx.x * 32 + (threadIdx.x >> 3), c = threadIdx.x & 7, n = m.n_dof, d = m.d;
    if (t >= B || c >= n) return;
    const int h = slab > 0 ? t / slab : 0;   // slab > 0: row t is rollout t - h*slab of state slab h ([n][ldq] each)
    const float q = qT[((size_t)h * n + c) * ldq + (t - h * slab)];
    const float sq = omds_sinf(q), cq = omds_cosf(q);
    Fq[(size_t)t * OMDS_FROW + c] = q;
    Fq[(size_t)t * OMDS_FROW + d + c] = sq;
    Fq[(size_t)t * OMDS_FROW + 2 * d + c] = cq;
    if (FqH) {   // the screening kernel's input row of this rollout (the other slots stay zero)
        FqH[omds_screen_fidx(c, t, ldF)] = (_Float16)q;
        FqH[omds_screen_fidx(d + c, t, ldF)] = (_Float16)sq;
        FqH[omds_screen_fidx(2 * d + c, t, ldF)] = (_Float16)cq;
        if (m.scrQ) {   // skip-connection networks: the same values at the slots of the concatenated columns
            _Float16* S = reinterpret_cast<_Float16*>(m.scrQ);
            S[omds_screen_sidx(c, 3 * d, t, ldF)] = (_Float16)q;
            S[omds_screen_sidx(d + c, 3 * d, t, ldF)] = (_Float16)sq;
            S[omds_screen_sidx(2 * d + c, 3 * d, t, ldF)] = (_Float16)cq;
        }
    }
}

__global__ __launch_bounds__(256) void k_obstacle_features(MlpDev m, const float* __restrict__ xyzr, int O,
                                                           float* __restrict__ Fp, float* __restrict__ radius,
                                                           _Float16* __restrict__ FpH, int ldF) {
    const int o = blockIdx.x * 64 + (threadIdx.x >> 2), c = threadIdx.x & 3, n = m.n_dof, d = m.d, po = d - n;   // po = 3 (x, y, z) or 2 (toy networks)
    if (o >= O) return;
    if (c == 3) { radius[o] = xyzr[o * 4 + 3]; return; }
    if (c >= po) return;
    const float p = xyzr[o * 4 + c];
    const float sp = omds_sinf(p), cp = omds_cosf(p);
    Fp[(size_t)o * OMDS_FROW + n + c] = p;
    Fp[(size_t)o * OMDS_FROW + d + n + c] = sp;
    Fp[(size_t)o * OMDS_FROW + 2 * d + n + c] = cp;
    if (FpH) {
        FpH[omds_screen_fidx(n + c, o, ldF)] = (_Float16)p;
        FpH[omds_screen_fidx(d + n + c, o, ldF)] = (_Float16)sp;
        FpH[omds_screen_fidx(2 * d + n + c, o, ldF)] = (_Float16)cp;
        if (m.scrP) {
            _Float16* S = reinterpret_cast<_Float16*>(m.scrP);
            S[omds_screen_sidx(n + c, 3 * d, o, ldF)] = (_Float16)p;
            S[omds_screen_sidx(d + n + c, 3 * d, o, ldF)] = (_Float16)sp;
            S[omds_screen_sidx(2 * d + n + c, 3 * d, o, ldF)] = (_Float16)cp;
        }
    }
}

template <int MT, int MR, int NR, int ACT>
__global__ __launch_bounds__((Geo<MT, MR, NR>::NT)) void k_pass1(MlpDev m, const float* __restrict__ Fq,
                                                               const float* __restrict__ Fp,
                                                               const float* __restrict__ radius, int O,
                                                               long long total_rows, uint32_t ignored,
                                                               float* __restrict__ Dmin, OmdsDivisor odiv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    pass1_tile<MT, MR, NR, ACT>(m, smem, Fq, Fp, radius, O, total_rows, ignored, Dmin, (long long)blockIdx.x * MT, odiv);
}

// Mixed-granularity launch: the first n_big workgroups take 64-row tiles, the rest cover the remaining rows
// in 32-row tiles.  Workgroups are dispatched in index order, so the kernel ends on small tiles and the
// drain phase (slots idling while the last tiles finish) shrinks with the tile size.
template <int ACT>
__global__ __launch_bounds__(512) void k_pass1_mixed(const float* __restrict__ Fq, const float* __restrict__ Fp,
                                                     const float* __restrict__ radius, float* __restrict__ Dmin,
                                                     long long total_rows, int O, uint32_t ignored, int n_big,
                                                     OmdsDivisor odiv, MlpDev m) {
    // argument order: the scalars and pointers the layer-1 build needs come first so that they can be preloaded into
    // SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count); the weight-pack descriptor is fetched behind them
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < n_big) {
        pass1_tile<64, 2, 1, ACT>(m, smem, Fq, Fp, radius, O, total_rows, ignored, Dmin, (long long)b * 64, odiv);
    } else {
        pass1_tile<32, 1, 1, ACT>(m, smem, Fq, Fp, radius, O, total_rows, ignored, Dmin,
                                  (long long)n_big * 64 + (long long)(b - n_big) * 32, odiv);
    }
}

// k_pass1 on per-tile compacted levels (pass1_tile_dyn): the same launch shape as k_pass1_mixed
__global__ __launch_bounds__(512) void k_pass1_dyn(const float* __restrict__ Fq, const float* __restrict__ Fp,
                                                   const float* __restrict__ radius, float* __restrict__ Dmin,
                                                   long long total_rows, int O, uint32_t ignored, int n_big,
                                                   OmdsDivisor odiv, MlpDev m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < n_big) pass1_tile_dyn<64, 2>(m, smem, Fq, Fp, radius, O, total_rows, ignored, Dmin, (long long)b * 64, odiv);
    else pass1_tile_dyn<32, 1>(m, smem, Fq, Fp, radius, O, total_rows, ignored, Dmin, (long long)n_big * 64 + (long long)(b - n_big) * 32, odiv);
}

// The same two kernels in pass1_tile's MODE 6 (kernels of their own: the tuned mode-0 kernels keep their argument lists and code):
// beside Dmin every pair's pass-2 distance, arg-min link and ReLU masks go to `ex`, indexed by the pair.
template <int MT, int MR, int NR, int ACT>
__global__ __launch_bounds__((Geo<MT, MR, NR>::NT)) void k_pass1e(MlpDev m, const float* __restrict__ Fq,
                                                                const float* __restrict__ Fp,
                                                                const float* __restrict__ radius, int O,
                                                                long long total_rows, uint32_t ignored,
                                                                float* __restrict__ Dmin, OmdsDivisor odiv, ExactOut ex) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    pass1_tile<MT, MR, NR, ACT, 6>(m, smem, Fq, Fp, radius, O, total_rows, ignored, Dmin, (long long)blockIdx.x * MT, odiv, nullptr, nullptr, &ex);
}
template <int ACT>
__global__ __launch_bounds__(512) void k_pass1e_mixed(const float* __restrict__ Fq, const float* __restrict__ Fp,
                                                      const float* __restrict__ radius, float* __restrict__ Dmin,
                                                      long long total_rows, int O, uint32_t ignored, int n_big,
                                                      OmdsDivisor odiv, MlpDev m, ExactOut ex) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < n_big) {
        pass1_tile<64, 2, 1, ACT, 6>(m, smem, Fq, Fp, radius, O, total_rows, ignored, Dmin, (long long)b * 64, odiv, nullptr, nullptr, &ex);
    } else {
        pass1_tile<32, 1, 1, ACT, 6>(m, smem, Fq, Fp, radius, O, total_rows, ignored, Dmin,
                                     (long long)n_big * 64 + (long long)(b - n_big) * 32, odiv, nullptr, nullptr, &ex);
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
template <int ACT, int ROWS>
__global__ __launch_bounds__(P2_NT) void k_pass2(MlpDev m, const float* __restrict__ Fq,
                                                 const float* __restrict__ Fp, const float* __restrict__ radius,
                                                 const float* __restrict__ xyzr, const int32_t* __restrict__ idx,
                                                 int total_rows, int k, const float* __restrict__ qT, int ldq,
                                                 float* __restrict__ gradx, float* __restrict__ drow,
                                                 float* __restrict__ yraw, int32_t* __restrict__ minidx,
                                                 float* __restrict__ dscr, int seed_col) {
    // dscr (tanh only): [nhh+1][rows padded to 32][256] activation derivatives 1 - h^2, L2-resident scratch
    extern __shared__ __attribute__((aligned(16))) float smem[];
    P2Smem sm;
    sm.Hs = smem;
    sm.gf = sm.Hs + P2_MT * LDH;
    sm.maskL = reinterpret_cast<uint16_t*>(sm.gf + 32 * 33);
    sm.rowT = reinterpret_cast<int*>(sm.maskL + ((m.nhh + 2) / 2 * 2) * P2_NT);
    sm.rowO = sm.rowT + P2_MT;
    sm.rowMin = sm.rowO + P2_MT;
    const int tid = threadIdx.x;
    const int R0 = blockIdx.x * ROWS;
    if (tid < ROWS) {
        const int R = R0 + tid;
        int t = -1, o = 0;
        if (R < total_rows) { t = R / k; o = idx[R]; }
        sm.rowT[tid] = t;
        sm.rowO[tid] = o;
    }
    __syncthreads();
    pass2_body<ACT, ROWS>(m, sm, Fq, Fp, radius, xyzr, R0, total_rows, qT, ldq, gradx, drow, R0, yraw, minidx, dscr,
                                 (size_t)gridDim.x * ROWS * OMDS_WIDTH, R0, 0, nullptr, 0, seed_col);
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
void omds_launch_rollout_features(hipStream_t s, const MlpDev& m, const float* qT, int ldq, int B, float* Fq, uint16_t* FqH, int ldF, int slab) {
    if (B <= 0) return;
    hipLaunchKernelGGL(k_rollout_features, dim3((B + 31) / 32), dim3(256), 0, s, m, qT, ldq, B, Fq, reinterpret_cast<_Float16*>(FqH), ldF, slab);
}

void omds_launch_obstacle_features(hipStream_t s, const MlpDev& m, const float* xyzr, int O, float* Fp, float* radius, uint16_t* FpH, int ldF) {
    if (O <= 0) return;
    hipLaunchKernelGGL(k_obstacle_features, dim3((O + 63) / 64), dim3(256), 0, s, m, xyzr, O, Fp, radius, reinterpret_cast<_Float16*>(FpH), ldF);
}

template <int MT, int MR, int NR, int ACT>
static void launch_pass1_a(hipStream_t s, const MlpDev& m, const float* Fq, const float* Fp, const float* radius,
                           int O, long long total, uint32_t ignored, float* Dmin) {
    using G = Geo<MT, MR, NR>;
    const size_t lds = (size_t)MT * LDH * 4 + (size_t)MT * 4;
    static std::atomic<uint64_t> configured{0};
    if (omds_first_use_on_device(configured)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass1<MT, MR, NR, ACT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const long long tiles = (total + MT - 1) / MT;
    hipLaunchKernelGGL((k_pass1<MT, MR, NR, ACT>), dim3((unsigned)tiles), dim3(G::NT), lds, s, m, Fq, Fp, radius, O,
                       total, ignored, Dmin, OmdsDivisor::make((unsigned)O));
}

template <int MT, int MR, int NR>
static void launch_pass1_t(hipStream_t s, const MlpDev& m, const float* Fq, const float* Fp, const float* radius,
                           int O, long long total, uint32_t ignored, float* Dmin) {
    if (m.act == OMDS_ACT_RELU) launch_pass1_a<MT, MR, NR, OMDS_ACT_RELU>(s, m, Fq, Fp, radius, O, total, ignored, Dmin);
    else launch_pass1_a<MT, MR, NR, OMDS_ACT_TANH>(s, m, Fq, Fp, radius, O, total, ignored, Dmin);
}

// Tile choice: 64-row tiles (8 waves, each 64 rows x 32 columns, two workgroups per CU) once there are
// enough tiles to fill 256 CUs; 32-row tiles for small batches (planar configs, dist_grad calls).
template <int ACT>
static void launch_pass1_mixed(hipStream_t s, const MlpDev& m, const float* Fq, const float* Fp, const float* radius,
                               int O, long long total, uint32_t ignored, float* Dmin, int small_rounds) {
    const size_t lds = (size_t)64 * LDH * 4 + 64 * 4;
    static std::atomic<uint64_t> configured{0};
    if (omds_first_use_on_device(configured)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass1_mixed<ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    // keep `small_rounds` x 512 x 64 rows (in units of resident 64-row workgroups) for the 32-row tail tiles
    long long tiles64 = total / 64;
    long long keep = (long long)small_rounds * 512 / 2;   // 64-row tiles' worth of rows given to small tiles
    long long n_big = tiles64 > keep ? tiles64 - keep : 0;
    const long long rest = total - n_big * 64;
    const long long n_small = (rest + 31) / 32;
    hipLaunchKernelGGL((k_pass1_mixed<ACT>), dim3((unsigned)(n_big + n_small)), dim3(512), lds, s, Fq, Fp, radius, Dmin,
                       total, O, ignored, (int)n_big, OmdsDivisor::make((unsigned)O), m);
}

void omds_launch_pass1(hipStream_t s, const MlpDev& m, const float* Fq, const float* Fp, const float* radius,
                       int O, int B, uint32_t ignored, float* Dmin) {
    const long long total = (long long)B * O;
    if (total <= 0) return;
    static const int forced = OMDS_EXP_ENV("OMDS_PASS1_VARIANT", 0);   // experiment builds: a tile variant instead of the rule below
    int v = forced;
    // measured on MI355X (profiles/): 64-row tiles (2 workgroups per CU) beat 128-row tiles at N*O = 301k rows
    // (129 vs 123 TFLOP/s: shorter tail) and tie at 1.2M rows (134 TFLOP/s)
    // and ending the launch on one "round" of 32-row tiles (variant 11) shortens the drain: 135 vs 133 TFLOP/s
    // tiny batches (<= 128 32-row tiles: integrator tick, planar toy shapes): 16-row tiles on the 16x16x4 MFMA halve the
    // chain of dependent GEMMs that is the whole latency of such a launch
    if (v == 0) v = (total >= 64LL * 1024) ? 11 : ((total >= 64LL * 512) ? 3 : (total <= 32LL * 128 ? 6 : 5));
    if (m.compact && v != 6) {   // per-tile compaction: 64-row tiles, the last round (or everything, for small batches) in 32-row tiles
        static const int lds_pad = OMDS_EXP_ENV("OMDS_DYN_LDS_PAD", 0);   // experiment builds: bytes of unused LDS (one workgroup per CU: 20000)
        const size_t lds = (size_t)64 * LDH * 4 + 64 * 4 + 32 + (size_t)OMDS_IDS * 4 + (size_t)lds_pad;
        static std::atomic<uint64_t> configured{0};
        if (omds_first_use_on_device(configured))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass1_dyn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const long long tiles64 = total / 64, keep = v >= 10 ? (long long)(v - 10) * 512 / 2 : (v == 5 ? tiles64 : 0);
        const long long n_big = tiles64 > keep ? tiles64 - keep : 0;
        const long long n_small = (total - n_big * 64 + 31) / 32;
        hipLaunchKernelGGL(k_pass1_dyn, dim3((unsigned)(n_big + n_small)), dim3(512), lds, s, Fq, Fp, radius, Dmin, total, O, ignored, (int)n_big,
                           OmdsDivisor::make((unsigned)O), m);
        return;
    }
    if (v >= 10) {   // 10 + r: mixed tiles, the last r "rounds" of 512 workgroups use 32-row tiles
        if (m.act == OMDS_ACT_RELU) launch_pass1_mixed<OMDS_ACT_RELU>(s, m, Fq, Fp, radius, O, total, ignored, Dmin, v - 10);
        else launch_pass1_mixed<OMDS_ACT_TANH>(s, m, Fq, Fp, radius, O, total, ignored, Dmin, v - 10);
        return;
    }
    switch (v) {
        case 1: launch_pass1_t<128, 4, 1>(s, m, Fq, Fp, radius, O, total, ignored, Dmin); break;
        case 3: launch_pass1_t<64, 2, 1>(s, m, Fq, Fp, radius, O, total, ignored, Dmin); break;
        case 6: launch_pass1_t<16, 1, 1>(s, m, Fq, Fp, radius, O, total, ignored, Dmin); break;
        default: launch_pass1_t<32, 1, 1>(s, m, Fq, Fp, radius, O, total, ignored, Dmin); break;
    }
}

// pass 1 that also leaves every pair's pass-2 distance, arg-min link and ReLU masks (pass1_tile MODE 6) -- the same tile choice as
// omds_launch_pass1, so Dmin is the same launch shape's bits (they are the same bits in every shape anyway).  ReLU networks only.
template <int MT, int MR, int NR>
static void launch_pass1e_t(hipStream_t s, const MlpDev& m, const float* Fq, const float* Fp, const float* radius,
                            int O, long long total, uint32_t ignored, float* Dmin, const ExactOut& ex) {
    using G = Geo<MT, MR, NR>;
    const size_t lds = (size_t)MT * LDH * 4 + (size_t)MT * 8 + (size_t)MT * (m.nhh + 1) * 32;
    static std::atomic<uint64_t> configured{0};
    if (omds_first_use_on_device(configured)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass1e<MT, MR, NR, OMDS_ACT_RELU>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)MT * LDH * 4 + (size_t)MT * 8 + (size_t)MT * (OMDS_MAX_HIDDEN + 1) * 32));
    }
    const long long tiles = (total + MT - 1) / MT;
    hipLaunchKernelGGL((k_pass1e<MT, MR, NR, OMDS_ACT_RELU>), dim3((unsigned)tiles), dim3(G::NT), lds, s, m, Fq, Fp, radius, O,
                       total, ignored, Dmin, OmdsDivisor::make((unsigned)O), ex);
}
void omds_launch_pass1_emit(hipStream_t s, const MlpDev& m, const float* Fq, const float* Fp, const float* radius,
                            int O, int B, uint32_t ignored, float* Dmin, const ExactOut& ex) {
    const long long total = (long long)B * O;
    if (total <= 0) return;
    if (total >= 64LL * 1024) {
        const size_t lds = (size_t)64 * LDH * 4 + 64 * 8 + (size_t)64 * (m.nhh + 1) * 32;
        static std::atomic<uint64_t> configured{0};
        if (omds_first_use_on_device(configured)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass1e_mixed<OMDS_ACT_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)((size_t)64 * LDH * 4 + 64 * 8 + (size_t)64 * (OMDS_MAX_HIDDEN + 1) * 32));
        }
        const long long tiles64 = total / 64, keep = 512 / 2;   // one round of 32-row tiles at the end, like omds_launch_pass1
        const long long n_big = tiles64 > keep ? tiles64 - keep : 0;
        const long long n_small = (total - n_big * 64 + 31) / 32;
        hipLaunchKernelGGL((k_pass1e_mixed<OMDS_ACT_RELU>), dim3((unsigned)(n_big + n_small)), dim3(512), lds, s, Fq, Fp, radius, Dmin,
                           total, O, ignored, (int)n_big, OmdsDivisor::make((unsigned)O), m, ex);
    } else if (total >= 64LL * 512) launch_pass1e_t<64, 2, 1>(s, m, Fq, Fp, radius, O, total, ignored, Dmin, ex);
    else if (total <= 32LL * 128) launch_pass1e_t<16, 1, 1>(s, m, Fq, Fp, radius, O, total, ignored, Dmin, ex);
    else launch_pass1e_t<32, 1, 1>(s, m, Fq, Fp, radius, O, total, ignored, Dmin, ex);
}

void omds_launch_topk(hipStream_t s, const float* Dmin, int B, int O, int k, int32_t* idx) {
    if (B <= 0) return;
    hipLaunchKernelGGL(k_topk, dim3((B + 3) / 4), dim3(256), 0, s, Dmin, B, O, k, idx);
}

void omds_launch_pass2(hipStream_t s, const MlpDev& m, const float* Fq, const float* Fp, const float* radius,
                       const float* xyzr, const int32_t* idx, int B, int k, const float* qT, int ldq, float* gradx,
                       float* drow, float* yraw, int32_t* minidx, float* dscr, int seed_col) {
    const int total = B * k;
    if (total <= 0) return;
    const size_t lds = ((size_t)P2_MT * LDH + 32 * 33) * 4 + (size_t)(m.nhh + 1) * P2_NT * 4 + 3 * P2_MT * 4;
    const int maxlds = (int)(((size_t)P2_MT * LDH + 32 * 33) * 4 + (size_t)(OMDS_MAX_HIDDEN + 1) * P2_NT * 4 + 3 * P2_MT * 4);
    static std::atomic<uint64_t> configured{0};
    if (omds_first_use_on_device(configured)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass2<OMDS_ACT_RELU, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass2<OMDS_ACT_TANH, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass2<OMDS_ACT_RELU, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass2<OMDS_ACT_TANH, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
    }
    // same tile height as the fused tail would pick for this batch, so that the batch entry points (omds_dist_grad)
    // reproduce the step path bit for bit
    const int rows = omds_tail_rows(B, k);
    const dim3 grid((total + rows - 1) / rows);
#define OMDS_P2_LAUNCH(A, R) hipLaunchKernelGGL((k_pass2<A, R>), grid, dim3(P2_NT), lds, s, m, Fq, Fp, radius, xyzr, idx, total, k, qT, ldq, gradx, drow, yraw, minidx, dscr, seed_col)
    if (m.act == OMDS_ACT_RELU) { if (rows == 16) OMDS_P2_LAUNCH(OMDS_ACT_RELU, 16); else OMDS_P2_LAUNCH(OMDS_ACT_RELU, 32); }
    else { if (rows == 16) OMDS_P2_LAUNCH(OMDS_ACT_TANH, 16); else OMDS_P2_LAUNCH(OMDS_ACT_TANH, 32); }
#undef OMDS_P2_LAUNCH
}

void omds_launch_blend(hipStream_t s, const float* gradx, const float* drow, int B, int k, int d, int n,
                       float softmax_k, float* dist, float* nngrad) {
    if (B <= 0) return;
    hipLaunchKernelGGL(k_blend, dim3((B + 255) / 256), dim3(256), 0, s, gradx, drow, B, k, d, n, softmax_k, dist, nngrad);
}

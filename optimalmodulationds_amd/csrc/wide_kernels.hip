// Distance networks wider than the fused kernels' 256 columns (gfx950).
//
// MLPRegression is width-agnostic (network_macros_mod.py:96-135); every network the reference ships is 256 (or 128) wide and runs on
// the fused LDS-resident MFMA kernels of this library.  A network with a wider hidden layer (<= 4096) takes THIS path: the
// reference's own unfused op sequence (MPPI.py:227-282, robot_sdf.py:153-158) on the exact-fp32 MFMA GEMM of the trainer
// (train.hip: k_gemm, 128 x 128 tiles, operands through LDS), with the activations of a layer materialised in HBM -- 288 GB make
// that affordable, and it is the honest shape for a layer that does not fit a CU's LDS tile:
//   pass 1   chunks of <= 65536 (rollout, obstacle) pairs: [x, sin x, cos x] -> Linear + act ... -> Linear (GEMM per layer, bias +
//            activation fused into the stores) -> min over the un-ignored links of y / out_div - radius -> Dmin
//   top-k    k_topk (mlp_kernels.hip), unchanged
//   pass 2   the N*k selected rows: the same forward with every layer's activation kept, arg-min over ALL raw outputs, seed =
//            W_last[arg-min] * act'(h), input-gradient GEMMs (x act' fused), the first layer's gradient and the
//            positional-encoding chain rule -> gradx, drow for k_modulate / k_blend
// No screening, no fused tail, no skip concatenations on this path (omds_set_mlp_ex says so).
#include <algorithm>

#include "omds_internal.h"
#include "trig_device.h"

namespace {

// X[r][3 d] = [x, sin x, cos x] of pair row r (row0 + r = t * O + o): x = (q_t, obstacle point o)
__global__ __launch_bounds__(256) void k_wide_encode_pairs(const float* __restrict__ qT, int ldq, const float* __restrict__ xyzr, long long row0,
                                                           int nrows, int O, int n, int d, float* __restrict__ X) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)nrows * d) return;
    const int r = (int)(e / d), j = (int)(e - (long long)r * d);
    const long long pair = row0 + r;
    const int t = (int)(pair / O), o = (int)(pair - (long long)t * O);
    const float x = j < n ? qT[(size_t)j * ldq + t] : xyzr[(size_t)o * 4 + (j - n)];
    float* xr = X + (size_t)r * 3 * d;
    xr[j] = x;
    xr[d + j] = omds_sinf(x);
    xr[2 * d + j] = omds_cosf(x);
}

// pass-2 rows: row r = (rollout r / k, its j-th closest obstacle idx[r]); also keeps the obstacle of each row
__global__ __launch_bounds__(256) void k_wide_encode_sel(const float* __restrict__ qT, int ldq, const float* __restrict__ xyzr,
                                                         const int32_t* __restrict__ idx, int rows, int k, int n, int d, float* __restrict__ X) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)rows * d) return;
    const int r = (int)(e / d), j = (int)(e - (long long)r * d);
    const int t = r / k, o = idx[r];
    const float x = j < n ? qT[(size_t)j * ldq + t] : xyzr[(size_t)o * 4 + (j - n)];
    float* xr = X + (size_t)r * 3 * d;
    xr[j] = x;
    xr[d + j] = omds_sinf(x);
    xr[2 * d + j] = omds_cosf(x);
}

// raw rows x [B][d] (omds_mlp_forward_vjp): the same encoding
__global__ __launch_bounds__(256) void k_wide_encode_raw(const float* __restrict__ x, int rows, int d, float* __restrict__ X) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)rows * d) return;
    const int r = (int)(e / d), j = (int)(e - (long long)r * d);
    const float v = x[e];
    float* xr = X + (size_t)r * 3 * d;
    xr[j] = v;
    xr[d + j] = omds_sinf(v);
    xr[2 * d + j] = omds_cosf(v);
}

// Dmin[row0 + r] = min over the un-ignored links of y / out_div - radius(o)   (MPPI.py:236-242)
__global__ __launch_bounds__(256) void k_wide_min(const float* __restrict__ Y, const float* __restrict__ xyzr, long long row0, int nrows, int O,
                                                  int C, uint32_t ignored, float out_div, float* __restrict__ Dmin) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const long long pair = row0 + r;
    const int o = (int)(pair % O);
    const float rad = xyzr[(size_t)o * 4 + 3];
    float m = __builtin_inff();
    for (int c = 0; c < C; ++c) {
        const float v = ((ignored >> c) & 1u) ? 1e6f : Y[(size_t)r * C + c] / out_div - rad;
        m = fminf(m, v);
    }
    Dmin[pair] = m;
}

// arg-min over ALL raw outputs (robot_sdf.py:155), the distance of that link, and the backward seed
// G[r][c] = W_last[arg-min][c] * act'(h_last[r][c])
__global__ __launch_bounds__(256) void k_wide_seed(const float* __restrict__ Y, const float* __restrict__ Wlast, const float* __restrict__ Hlast,
                                                   const float* __restrict__ xyzr, const int32_t* __restrict__ obs_of_row, int rows, int C, int width,
                                                   int act, float out_div, float* __restrict__ drow, int32_t* __restrict__ minidx,
                                                   float* __restrict__ yraw, float* __restrict__ G, int seed_col) {
    const int r = blockIdx.x;
    if (r >= rows) return;
    __shared__ int s_min;
    if (threadIdx.x == 0) {
        int bi = 0;
        float bv = Y[(size_t)r * C];
        for (int c = 1; c < C; ++c) { const float v = Y[(size_t)r * C + c]; if (v < bv) { bv = v; bi = c; } }
        if (seed_col >= 0) { bi = seed_col; bv = Y[(size_t)r * C + seed_col]; }   // one Jacobian column (robot_sdf.py:92-100)
        s_min = bi;
        const float rad = obs_of_row ? xyzr[(size_t)obs_of_row[r] * 4 + 3] : 0.f;
        drow[r] = bv / out_div - rad;
        if (minidx) minidx[r] = bi;
    }
    if (yraw) for (int c = threadIdx.x; c < OMDS_CPAD; c += blockDim.x) yraw[(size_t)r * OMDS_CPAD + c] = c < C ? Y[(size_t)r * C + c] : 0.f;
    __syncthreads();
    const float* w = Wlast + (size_t)s_min * width;
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
        const float h = Hlast[(size_t)r * width + c];
        G[(size_t)r * width + c] = w[c] * (act == OMDS_ACT_RELU ? (h > 0.f ? 1.f : 0.f) : 1.f - h * h);
    }
}

// positional-encoding chain rule: d/dx = gf[:d] + gf[d:2d] * cos x - gf[2d:] * sin x   (x, sin x, cos x kept in X)
__global__ __launch_bounds__(256) void k_wide_chain(const float* __restrict__ gf, const float* __restrict__ X, int rows, int d, float* __restrict__ gradx) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)rows * d) return;
    const int r = (int)(e / d), j = (int)(e - (long long)r * d);
    const float* g = gf + (size_t)r * 3 * d;
    const float* xr = X + (size_t)r * 3 * d;
    gradx[e] = g[j] + g[d + j] * xr[2 * d + j] - g[2 * d + j] * xr[d + j];
}

}  // namespace

static inline unsigned blocks_of(long long n) { return (unsigned)((n + 255) / 256); }

// pass 1 over B rollouts x O obstacles -> Dmin [B][O]
int omds_wide_pass1(omds_ctx* ctx, const float* qT, int ldq, int B) {
    WideNet& w = ctx->wide;
    hipStream_t s = ctx->stream;
    const int O = ctx->n_obs, n = ctx->cfg.n_dof, d = w.d, L = (int)w.dims.size() - 1, C = w.dims[L];
    const long long total = (long long)B * O;
    for (long long row0 = 0; row0 < total; row0 += w.chunk_rows) {
        const int rows = (int)std::min<long long>(w.chunk_rows, total - row0);
        hipLaunchKernelGGL(k_wide_encode_pairs, dim3(blocks_of((long long)rows * d)), dim3(256), 0, s, qT, ldq, ctx->d_obs, row0, rows, O, n, d, w.X);
        const float* in = w.X;
        for (int i = 0; i < L; ++i) {
            float* out = w.H[i & 1];
            omds_launch_linear_forward(s, in, w.dims[i], w.W[i], w.b[i], out, w.dims[i + 1], rows, i + 1 < L ? w.act : -1);
            in = out;
        }
        hipLaunchKernelGGL(k_wide_min, dim3(blocks_of(rows)), dim3(256), 0, s, in, ctx->d_obs, row0, rows, O, C, ctx->prm.ignored_links, w.out_div, ctx->d_Dmin);
    }
    return OMDS_OK;
}

// pass 2 on `rows` rows whose encoded inputs are in w.X2 (obs_of_row: their obstacles, or nullptr for radius 0) -> gradx [rows][d], drow
int omds_wide_pass2_rows(omds_ctx* ctx, int rows, const int32_t* obs_of_row, float* gradx, float* drow, float* yraw, int32_t* minidx, int seed_col = -1) {
    WideNet& w = ctx->wide;
    hipStream_t s = ctx->stream;
    const int d = w.d, L = (int)w.dims.size() - 1, C = w.dims[L];
    const float* in = w.X2;
    for (int i = 0; i < L; ++i) {   // forward, every layer's activation kept (A[i] = output of Linear i)
        omds_launch_linear_forward(s, in, w.dims[i], w.W[i], w.b[i], w.A[i], w.dims[i + 1], rows, i + 1 < L ? w.act : -1);
        in = w.A[i];
    }
    const int wl = w.dims[L - 1];
    hipLaunchKernelGGL(k_wide_seed, dim3(rows), dim3(256), 0, s, w.A[L - 1], w.W[L - 1], w.A[L - 2], ctx->d_obs, obs_of_row, rows, C, wl, w.act, w.out_div,
                       drow, minidx, yraw, w.G[0], seed_col);
    float* G = w.G[0];
    float* Gn = w.G[1];
    for (int i = L - 2; i >= 1; --i) {   // gradient at the output of Linear i-1 through Linear i, times act'(h_{i-1})
        omds_launch_linear_inputgrad(s, G, w.dims[i + 1], w.W[i], w.dims[i], Gn, rows, w.A[i - 1], w.act);
        std::swap(G, Gn);
    }
    omds_launch_linear_inputgrad(s, G, w.dims[1], w.W[0], w.dims[0], Gn, rows, nullptr, -1);   // gradient at the encoded input
    hipLaunchKernelGGL(k_wide_chain, dim3(blocks_of((long long)rows * d)), dim3(256), 0, s, Gn, w.X2, rows, d, gradx);
    (void)C;
    return OMDS_OK;
}

// the whole network step of omds_propagate / omds_dist_grad on B states: Dmin, idx, gradx, drow
int omds_wide_network(omds_ctx* ctx, const float* qT, int ldq, int B) {
    WideNet& w = ctx->wide;
    const int O = ctx->n_obs, k = ctx->cfg.n_closest, n = ctx->cfg.n_dof;
    int rc;
    if ((rc = omds_wide_pass1(ctx, qT, ldq, B))) return rc;
    omds_launch_topk(ctx->stream, ctx->d_Dmin, B, O, k, ctx->d_idx);
    const int rows = B * k;
    hipLaunchKernelGGL(k_wide_encode_sel, dim3(blocks_of((long long)rows * w.d)), dim3(256), 0, ctx->stream, qT, ldq, ctx->d_obs, ctx->d_idx, rows, k, n, w.d, w.X2);
    return omds_wide_pass2_rows(ctx, rows, ctx->d_idx, ctx->d_gradx, ctx->d_drow, nullptr, nullptr);
}

// omds_mlp_forward_vjp on raw rows x [rows][d] (device pointer)
int omds_wide_vjp(omds_ctx* ctx, const float* d_x, int rows, int seed_col) {
    WideNet& w = ctx->wide;
    hipLaunchKernelGGL(k_wide_encode_raw, dim3(blocks_of((long long)rows * w.d)), dim3(256), 0, ctx->stream, d_x, rows, w.d, w.X2);
    return omds_wide_pass2_rows(ctx, rows, nullptr, ctx->d_gradx, ctx->d_drow, ctx->d_yraw, ctx->d_minidx, seed_col);
}

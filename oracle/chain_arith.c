/* CPU ORACLE, C part: the reference's fp32 ARITHMETIC restated.  TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg load it through oracle/chain.py; the package optimalmodulationds_amd never does).
 *
 * What torch-CPU computes for the reference's network (python_scripts/mlp_learn/sdf/network_macros_mod.py:137-146,
 * robot_sdf.py:153-158), established bit for bit in this container (tools/studies/assoc_order_study.py --torch,
 * profiles/r06_assoc_order_study.txt; torch 2.10.0, MKL 2024.2, AVX-512):
 *   - nn.Linear = addmm(bias, x, W^T) = MKL sgemm: for M >= 11 rows every output element is ONE fmaf chain over k in ASCENDING
 *     order starting from ZERO, and the bias is added AFTER the product            -> omds_orc_linear
 *   - the vjp's products (g * mask) @ W are the same chains over the layer's output units -> omds_orc_matmul
 *   - torch.sin / torch.cos are MKL VML's vmsSin / vmsCos (HA mode), a closed implementation; the nearest PUBLISHED algorithm is
 *     SLEEF's 1.0-ULP xsinf_u1 / xcosf_u1 (98.0 % / 97.0 % of 1e8 inputs bit-identical to torch's, one ulp otherwise; a correctly
 *     rounded sine agrees on 95 %), restated here in its FMA form for |x| < 125                -> omds_orc_sin / omds_orc_cos
 * Built by oracle/chain.py (and __graft_entry__.build()):  gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp -shared -fPIC
 * (-ffp-contract=off: only the fmaf calls written below fuse). */
#include <math.h>
/* a FIXED small team, and only for big batches: the GPU boxes have 128+ cores shared with other jobs, where a default-sized OpenMP team
 * spins for longer than these loops run (tests/test_gpu_parity.py went from 150 s to 600 s on a busy box) */
#define OMDS_ORC_THREADS 8
#include <stdlib.h>
#include <string.h>

/* y[m][n] = (sum over k ascending, one fmaf chain from 0, of x[m][k] W[n][k]) + b[n];  W [N][K] like torch, b may be NULL */
void omds_orc_linear(const float* x, const float* W, const float* b, float* y, long M, int K, int N) {
    float* Wt = (float*)malloc((size_t)K * N * sizeof(float));   /* [K][N]: the chain of column n reads Wt[k][n] */
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) Wt[(size_t)k * N + n] = W[(size_t)n * K + k];
#pragma omp parallel for schedule(static) num_threads(OMDS_ORC_THREADS) if (M >= 1024)
    for (long m = 0; m < M; ++m) {
        float* acc = y + (size_t)m * N;
        for (int n = 0; n < N; ++n) acc[n] = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float xv = x[(size_t)m * K + k];
            const float* w = Wt + (size_t)k * N;
            for (int n = 0; n < N; ++n) acc[n] = fmaf(xv, w[n], acc[n]);
        }
        if (b)
            for (int n = 0; n < N; ++n) acc[n] = acc[n] + b[n];
    }
    free(Wt);
}

/* out[m][n] = sum over k ascending (one fmaf chain from 0) of g[m][k] W[k][n];  W [K][N] row-major (torch: g @ W) */
void omds_orc_matmul(const float* g, const float* W, float* out, long M, int K, int N) {
#pragma omp parallel for schedule(static) num_threads(OMDS_ORC_THREADS) if (M >= 1024)
    for (long m = 0; m < M; ++m) {
        float* acc = out + (size_t)m * N;
        for (int n = 0; n < N; ++n) acc[n] = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float gv = g[(size_t)m * K + k];
            const float* w = W + (size_t)k * N;
            for (int n = 0; n < N; ++n) acc[n] = fmaf(gv, w[n], acc[n]);
        }
    }
}

/* ---- SLEEF xsinf_u1 / xcosf_u1 (sleefsimdsp.c), |d| < 125, double-float helpers of df.h in their FMA form ---- */
typedef struct { float x, y; } f2;
static f2 add2_ff(float x, float y) { f2 r; r.x = x + y; float v = r.x - x; r.y = (x - (r.x - v)) + (y - v); return r; }
static f2 add2_f2f(f2 x, float y) { f2 r; r.x = x.x + y; float v = r.x - x.x; float w = (x.x - (r.x - v)) + (y - v); r.y = x.y + w; return r; }
static f2 add_f2f(f2 x, float y) { f2 r; r.x = x.x + y; r.y = x.y + ((x.x - r.x) + y); return r; }
static f2 add_ff(float x, float y) { f2 r; r.x = x + y; r.y = (x - r.x) + y; return r; }
static f2 add_ff2(float x, f2 y) { f2 r; r.x = x + y.x; r.y = ((x - r.x) + y.x) + y.y; return r; }
static f2 squ(f2 x) { f2 r; r.x = x.x * x.x; r.y = fmaf(x.x + x.x, x.y, fmaf(x.x, x.x, -r.x)); return r; }
static f2 mul(f2 x, f2 y) { f2 r; r.x = x.x * y.x; r.y = fmaf(x.x, y.y, fmaf(x.y, y.x, fmaf(x.x, y.x, -r.x))); return r; }
static float mul_to_f(f2 x, f2 y) { float p = x.x * y.y; return fmaf(x.x, y.x, fmaf(x.y, y.x, p)); }
#define PI_A2 3.1414794921875f
#define PI_B2 0.00011315941810607910156f
#define PI_C2 1.9841872589410058936e-09f
#define M_1_PI_F 0.318309886183790671537767526745028724f
static float sin_reduced(f2 t) {
    f2 s = squ(t);
    float u = 2.6083159809786593541503e-06f;
    u = fmaf(u, s.x, -0.0001981069071916863322258f);
    u = fmaf(u, s.x, 0.00833307858556509017944336f);
    f2 x = add_ff2(1.0f, mul(add_ff(-0.166666597127914428710938f, u * s.x), s));
    return mul_to_f(t, x);
}
static float sin_u10(float d) {
    if (!(fabsf(d) < 125.0f)) return sinf(d);
    float u = rintf(d * M_1_PI_F);
    int q = (int)u;
    float v = fmaf(u, -PI_A2, d);
    f2 s = add2_ff(v, u * -PI_B2);
    s = add_f2f(s, u * -PI_C2);
    float r = sin_reduced(s);
    if (q & 1) r = -r;
    return (d == 0.0f && signbit(d)) ? d : r;
}
static float cos_u10(float d) {
    if (!(fabsf(d) < 125.0f)) return cosf(d);
    float dq = fmaf(rintf(fmaf(d, M_1_PI_F, -0.5f)), 2.0f, 1.0f);
    int q = (int)dq;
    f2 s = add2_ff(d, dq * (-PI_A2 * 0.5f));
    s = add2_f2f(s, dq * (-PI_B2 * 0.5f));
    s = add2_f2f(s, dq * (-PI_C2 * 0.5f));
    float r = sin_reduced(s);
    if ((q & 2) == 0) r = -r;
    return r;
}
void omds_orc_sin(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = sin_u10(x[i]); }
void omds_orc_cos(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = cos_u10(x[i]); }

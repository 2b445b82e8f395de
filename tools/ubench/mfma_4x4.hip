// v_mfma_f32_4x4x1_16B_f32 with the A operand of ONE block broadcast to all 16 blocks (cbsz = 4, abid = g): is it
//   D[i][l] = fmaf(A[4g + i], B[l], C[i][l])   for register i = 0..3 and lane l = 0..63
// i.e. 4 rows x 64 columns x k = 1 per instruction, one fused multiply-add per element?  If so, a GEMM on 4-row groups
// reproduces the k-ordered fmaf chains of the 32-row / 16-row kernels (tools/ubench/mfma_order.hip) with a row granularity of 4.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int G>
__global__ void k4(const float* a, const float* b, const float* c, float* out) {
    const int l = threadIdx.x;
    f32x4 acc = {c[l], c[64 + l], c[128 + l], c[192 + l]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 4, G, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[64 + l], b[64 + l], acc, 4, G, 0);   // second k step: a chain
    for (int i = 0; i < 4; ++i) out[i * 64 + l] = acc[i];
}

// Issue rate: NACC independent accumulator chains per wave, WPS waves per SIMD, n instructions per chain link.
template <int NACC>
__global__ void k_rate(float* out, unsigned long long* cyc, int iters) {
    const int l = threadIdx.x & 63;
    f32x4 acc[NACC];
    for (int g = 0; g < NACC; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = (float)l, b = 1.f / (1 + l);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int g = 0; g < NACC; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[g], 4, 3, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int g = 0; g < NACC; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    if (s == 12345.f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
void rate(float* out, unsigned long long* cyc, int waves_per_simd) {
    const int iters = 64;
    hipLaunchKernelGGL(k_rate<NACC>, dim3(1), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k_rate<NACC>, dim3(1), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("4x4x1 16B: %d chains, %d wave(s) per SIMD: %.2f cycles per MFMA of the SIMD\n", NACC, waves_per_simd, (double)c / (iters * 16.0 * NACC * waves_per_simd));
}

// Whole-chip rate by the wall clock: 256 workgroups x 8 waves, every wave n MFMAs on NACC chains
template <int SHAPE, int NACC>
__global__ __launch_bounds__(512) void k_chip(float* out, int iters) {
    const int l = threadIdx.x & 63;
    f32x4 acc[NACC];
    for (int g = 0; g < NACC; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = (float)l, b = 1.f / (1 + l);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int g = 0; g < NACC; ++g) {
                if (SHAPE == 4) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[g], 4, 3, 0);
                else acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int g = 0; g < NACC; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    if (s == 12345.f) out[0] = s;
}
template <int SHAPE, int NACC>
void chip(float* out, int threads) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chip<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_chip<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (SHAPE == 4 ? 512.0 : 2048.0) * iters * 16.0 * NACC * (threads / 64) * 256.0;
    printf("%s, %d chains, %d waves per CU on 256 CUs: %.1f TFLOP/s by the wall clock (%.3f ms)\n", SHAPE == 4 ? "4x4x1 16B " : "16x16x4   ", NACC, threads / 64, flop / ms * 1e-9, ms);
}

int main() {
    {
        float* o;
        hipMalloc(&o, 4);
        chip<16, 2>(o, 256); chip<16, 2>(o, 512); chip<16, 4>(o, 512);
        chip<4, 5>(o, 256); chip<4, 5>(o, 512); chip<4, 3>(o, 512); chip<4, 5>(o, 1024);
    }
    {
        float* o;
        unsigned long long* c;
        hipMalloc(&o, 4);
        hipMalloc(&c, 8);
        rate<1>(o, c, 1); rate<2>(o, c, 1); rate<3>(o, c, 1); rate<5>(o, c, 1); rate<2>(o, c, 2); rate<3>(o, c, 2); rate<5>(o, c, 2);
    }
    float *da, *db, *dc, *dout;
    hipMalloc(&da, 128 * 4); hipMalloc(&db, 128 * 4); hipMalloc(&dc, 256 * 4); hipMalloc(&dout, 256 * 4);
    unsigned st = 777;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((int)(st >> 8) - (1 << 23)) * (1.0f / (1 << 20)); };
    long ok[3] = {0, 0, 0}, total = 0;
    for (int t = 0; t < 300; ++t) {
        float a[128], b[128], c[256], o[256];
        for (int i = 0; i < 128; ++i) { a[i] = rnd(); b[i] = rnd(); }
        for (int i = 0; i < 256; ++i) c[i] = rnd() * 3.f;
        hipMemcpy(da, a, 512, hipMemcpyHostToDevice); hipMemcpy(db, b, 512, hipMemcpyHostToDevice); hipMemcpy(dc, c, 1024, hipMemcpyHostToDevice);
        const int g = t % 3 == 0 ? 0 : (t % 3 == 1 ? 5 : 15);
        if (g == 0) hipLaunchKernelGGL(k4<0>, dim3(1), dim3(64), 0, 0, da, db, dc, dout);
        else if (g == 5) hipLaunchKernelGGL(k4<5>, dim3(1), dim3(64), 0, 0, da, db, dc, dout);
        else hipLaunchKernelGGL(k4<15>, dim3(1), dim3(64), 0, 0, da, db, dc, dout);
        hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 4; ++i)
            for (int l = 0; l < 64; ++l) {
                const float a0 = a[4 * g + i], a1 = a[64 + 4 * g + i], b0 = b[l], b1 = b[64 + l], c0 = c[i * 64 + l];
                const float chain = fmaf(a1, b1, fmaf(a0, b0, c0));
                const float unfused = (a1 * b1) + ((a0 * b0) + c0);
                const float exact = (float)((double)c0 + (double)a0 * b0 + (double)a1 * b1);
                ok[0] += o[i * 64 + l] == chain; ok[1] += o[i * 64 + l] == unfused; ok[2] += o[i * 64 + l] == exact;
                ++total;
            }
    }
    printf("4x4x1 16B, cbsz=4, abid=g: of %ld elements equal to  fmaf chain (rows 4g+i, column = lane): %ld   unfused mul+add: %ld   single rounding of the sum: %ld\n",
           total, ok[0], ok[1], ok[2]);
    return 0;
}

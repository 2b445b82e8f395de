// Which clock does s_memtime (__builtin_readcyclecounter) count on gfx950, and what is the shader clock under load?
// Each kernel runs a fixed instruction stream; we report s_memtime ticks and 100 MHz wall_clock64 ticks of one wave per
// workgroup, for a light VALU loop and for back-to-back fp16 / fp32 MFMAs (whose cycle count per instruction is known).
// Note: with two fp16-MFMA waves per SIMD the older wave takes the pipe until it is done (its window = its own MFMAs x 32
// cycles, the kernel lasts twice as long: compare the host-event time); the fp32 waves alternate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void k(unsigned long long* out, int iters, float* sink) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)1.f; b[j] = (_Float16)0.5f; }
    float v = threadIdx.x;
    h8 ra[8], rb[8];
    {
        unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int q = 0; q < 8; ++q)
            for (int j = 0; j < 8; ++j) {
                st = st * 1664525u + 1013904223u; ra[q][j] = (_Float16)(((int)(st >> 20) - 2048) * (1.0f / 4096.0f));
                st = st * 1664525u + 1013904223u; rb[q][j] = (_Float16)(((int)(st >> 20) - 2048) * (1.0f / 4096.0f));
            }
    }
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 64; ++u) v = v * 1.0001f + 0.5f;
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.f, 0.5f, acc[i], 0, 0, 0);
        } else if (MODE == 4) {   // the same random lanes every time (no toggling between consecutive MFMAs)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[0], rb[0], acc[i], 0, 0, 0);
        } else if (MODE == 5) {   // fp32 MFMA on pseudo-random operands
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)ra[(u + i) & 7][u & 7], (float)rb[(u * 3 + i) & 7][i], acc[i], 0, 0, 0);
        } else {   // fp16 MFMA on pseudo-random operands (8 different register sets, per lane): realistic bit toggling
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[(u + i) & 7], rb[(u * 3 + i) & 7], acc[i], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = v;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
void run(const char* name, int iters, double cyc_per_iter_per_simd, unsigned long long* d, float* sink) {
    unsigned long long h[512];
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, d, iters, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, d, iters, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0, w = 0;
    for (int i = 0; i < 256; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
    c /= 256; w /= 256;
    const double us = w / 100.0;
    printf("(host events: %.0f us) ", ms * 1000.0);
    printf("%-28s %.0f us: s_memtime %.0f ticks = %.3f GHz", name, us, c, c / us / 1000.0);
    if (cyc_per_iter_per_simd > 0) printf(";  MFMA pipe cycles needed %.0f -> shader clock >= %.3f GHz", cyc_per_iter_per_simd * iters, cyc_per_iter_per_simd * iters / us / 1000.0);
    printf("\n");
}

int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 512 * 8); hipMalloc(&sink, 256 * 512 * 4);
    // 512 threads = 2 waves per SIMD; per iteration a wave issues 64 MFMAs -> 128 per SIMD
    run<0>("VALU only", 20000, 0, d, sink);
    run<1>("fp16 MFMA constants (short)", 2000, 64 * 32.0, d, sink);
    run<1>("fp16 MFMA constants (long)", 200000, 64 * 32.0, d, sink);
    run<2>("fp32 MFMA constants (long)", 100000, 128 * 64.0, d, sink);
    run<3>("fp16 MFMA random data (long)", 200000, 64 * 32.0, d, sink);
    run<4>("fp16 MFMA random, one set", 200000, 64 * 32.0, d, sink);
    run<5>("fp32 MFMA random data", 100000, 128 * 64.0, d, sink);
    run<0>("VALU only again", 20000, 0, d, sink);
    return 0;
}

// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 vs waves per SIMD and independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd, float* d) {
    const int threads = 64 * 4 * waves_per_simd;   // one workgroup per CU
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, d, 10, 1.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, d, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * NACC * waves_per_simd;
    const double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
    const double tf = 256.0 * 4 * mfma_per_simd * 4096 / (ms * 1e-3) / 1e12;
    printf("waves/SIMD %d  acc/wave %d : %.1f cycles per MFMA per SIMD (at 2.4 GHz), %.1f TFLOP/s\n", waves_per_simd, NACC, cyc, tf);
}

int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    for (int w : {1, 2, 4}) { run<1>(w, d); run<2>(w, d); run<4>(w, d); run<8>(w, d); }
    return 0;
}

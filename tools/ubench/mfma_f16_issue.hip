// Microbenchmark: issue rate of v_mfma_f32_32x32x16_f16 (registers only) vs waves per SIMD and independent accumulators
// per wave, with and without an LDS fragment read (1 KiB per wave) per MFMA or per two MFMAs -- the practical ceiling of
// the screening kernel's inner loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// MODE 0: operands in registers; 1: one ds_read_b128 A fragment per MFMA; 2: one per two MFMAs (shared by two accumulators)
template <int NACC, int MODE>
__global__ void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u;
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)1.f; b[j] = (_Float16)0.5f; }
    const unsigned char* lp = lds + (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < NACC; ++i) {
                    const h8 f = *reinterpret_cast<const h8*>(lp + ((it * 16 + u * NACC + i) & 63) * 1024);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, b, acc[i], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NACC; i += 2) {
                    const h8 f = *reinterpret_cast<const h8*>(lp + ((it * 16 + u * NACC + i) & 63) * 1024);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, b, acc[i], 0, 0, 0);
                    acc[i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, a, acc[i + 1], 0, 0, 0);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int MODE>
void run(int waves_per_simd, float* d) {
    const int threads = 64 * 4 * waves_per_simd;   // one workgroup per CU
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(256), dim3(threads), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 16 * NACC * waves_per_simd;
    const double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
    const double tf = 256.0 * 4 * mfma_per_simd * 32768 / (ms * 1e-3) / 1e12;
    printf("mode %d  waves/SIMD %d  acc/wave %d : %.1f cycles per MFMA per SIMD (at 2.4 GHz), %.0f TFLOP/s (%.0f %% of 2500)\n", MODE,
           waves_per_simd, NACC, cyc, tf, tf / 25.0);
}

int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    for (int w : {1, 2}) { run<1, 0>(w, d); run<2, 0>(w, d); run<4, 0>(w, d); }
    for (int w : {1, 2}) { run<1, 1>(w, d); run<2, 1>(w, d); run<4, 1>(w, d); }
    for (int w : {1, 2}) { run<2, 2>(w, d); run<4, 2>(w, d); }
    return 0;
}

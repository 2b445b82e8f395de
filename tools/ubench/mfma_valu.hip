// Microbenchmark: do packed-fp32 VALU FMAs execute in the shadow of an fp32 MFMA of the SAME wave?
// v_mfma_f32_32x32x2_f32 occupies the matrix pipe for 16 passes (64 cycles) but its issue takes only a few cycles; if
// independent v_pk_fma_f32 (2 FMAs x 64 lanes per instruction, same peak FLOP rate as the MFMA) can issue in between,
// a loop of 1 MFMA + n VALU FMAs still costs ~64 cycles per iteration for n up to ~14 and the SIMD does up to twice
// the fp32 work.  Prints cycles per iteration and the combined FLOP rate for n = 0..20 at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV>
__global__ void k(float* out, int iters, float a, float b) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    f32x2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = f32x2{(float)threadIdx.x, 1.f};
    const f32x2 x = {1.0001f, 0.9999f}, y = {a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i % 16]) : "v"(x), "v"(y));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + 8) % 16]) : "v"(x), "v"(y));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
void run(int wps, float* d) {
    const int threads = 64 * 4 * wps, iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(threads), 0, 0, d, 10, 1.f, 1.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(threads), 0, 0, d, iters, 1.f, 1.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 8 * wps;                  // MFMAs per SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / mf;
    const double tf_m = 256.0 * 4 * mf * 4096 / (ms * 1e-3) / 1e12;
    const double tf_v = 256.0 * 4 * mf * NV * 256 / (ms * 1e-3) / 1e12;
    printf("waves/SIMD %d  VALU pk_fma per MFMA %2d : %6.1f cycles per MFMA per SIMD, MFMA %6.1f + VALU %6.1f = %6.1f TFLOP/s\n", wps, NV, cyc,
           tf_m, tf_v, tf_m + tf_v);
}

int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    for (int w : {1, 2, 4}) { run<0>(w, d); run<2>(w, d); run<4>(w, d); run<8>(w, d); run<12>(w, d); run<14>(w, d); run<16>(w, d); }
    return 0;
}

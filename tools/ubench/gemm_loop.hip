// Microbenchmark of the pass-1 GEMM inner loop in isolation: one 64-row activation tile in LDS, packed weights
// streamed from L2, 8 waves per workgroup (64 rows x 32 columns each), WGS workgroups per CU.
#include "../../optimalmodulationds_amd/csrc/mlp_device.h"
#include <cstdio>
#include <vector>

template <int VAR>
__device__ __forceinline__ void gemm_var(const float* Hw, const float4* Wp, int cb0, int lane, f32x16 (&acc)[2][1]) {
    constexpr int MR = 2, NR = 1;
    const float* arow = Hw + (lane & 31) * LDH + 4 * (lane >> 5);
    const float4* wp = Wp + (size_t)cb0 * (32 * 64) + lane;
    float4 a0[MR], a1[MR], w0[NR], w1[NR];
    load_chunk<MR, NR>(arow, wp, 0, a0, w0);
#pragma unroll 1
    for (int c = 0; c < 32; c += 2) {
        if (VAR == 0) {          // shipped structure
            load_chunk<MR, NR>(arow, wp, c + 1, a1, w1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mfma_chunk<MR, NR>(a0, w0, acc);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk<MR, NR>(arow, wp, (c + 2) & 31, a0, w0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mfma_chunk<MR, NR>(a1, w1, acc);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (VAR == 1) {   // no priority flips
            load_chunk<MR, NR>(arow, wp, c + 1, a1, w1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk<MR, NR>(a0, w0, acc);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk<MR, NR>(arow, wp, (c + 2) & 31, a0, w0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk<MR, NR>(a1, w1, acc);
            __builtin_amdgcn_sched_barrier(0);
        } else if (VAR == 2) {   // MFMAs only (loads hoisted out): upper bound of the loop structure
            mfma_chunk<MR, NR>(a0, w0, acc);
            mfma_chunk<MR, NR>(a0, w0, acc);
        } else if (VAR == 3) {   // only the LDS reads
            load_a_only:
#pragma unroll
            for (int i = 0; i < MR; ++i) a1[i] = *reinterpret_cast<const float4*>(arow + i * 32 * LDH + 8 * (c + 1));
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk<MR, NR>(a0, w0, acc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MR; ++i) a0[i] = *reinterpret_cast<const float4*>(arow + i * 32 * LDH + 8 * ((c + 2) & 31));
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk<MR, NR>(a1, w0, acc);
            __builtin_amdgcn_sched_barrier(0);
        } else if (VAR == 6) {   // loads in the middle of the cluster
            mfma_chunk<MR, NR>(a0, w0, acc);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk<MR, NR>(arow, wp, (c + 2) & 31, a0, w0);
            load_chunk<MR, NR>(arow, wp, c + 1, a1, w1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk<MR, NR>(a1, w1, acc);
            __builtin_amdgcn_sched_barrier(0);
        } else if (VAR == 7) {   // no sched barriers at all (compiler schedule)
            load_chunk<MR, NR>(arow, wp, c + 1, a1, w1);
            mfma_chunk<MR, NR>(a0, w0, acc);
            load_chunk<MR, NR>(arow, wp, (c + 2) & 31, a0, w0);
            mfma_chunk<MR, NR>(a1, w1, acc);
        } else if (VAR == 8) {   // sched_group_barrier: 1 load group per 2 MFMAs
            load_chunk<MR, NR>(arow, wp, c + 1, a1, w1);
            mfma_chunk<MR, NR>(a0, w0, acc);
            __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk<MR, NR>(arow, wp, (c + 2) & 31, a0, w0);
            mfma_chunk<MR, NR>(a1, w1, acc);
            __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (VAR == 9) {   // interleave pattern M1 V1 M1 D1 M1 D1 M5
            load_chunk<MR, NR>(arow, wp, c + 1, a1, w1);
            mfma_chunk<MR, NR>(a0, w0, acc);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 5, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk<MR, NR>(arow, wp, (c + 2) & 31, a0, w0);
            mfma_chunk<MR, NR>(a1, w1, acc);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 5, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (VAR == 10) {   // interleave pattern M2 V1 M2 D1 M2 D1 M2
            load_chunk<MR, NR>(arow, wp, c + 1, a1, w1);
            mfma_chunk<MR, NR>(a0, w0, acc);
            __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk<MR, NR>(arow, wp, (c + 2) & 31, a0, w0);
            mfma_chunk<MR, NR>(a1, w1, acc);
            __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (VAR == 11) {   // interleave pattern M5 V1 M1 D1 M1 D1 M1
            load_chunk<MR, NR>(arow, wp, c + 1, a1, w1);
            mfma_chunk<MR, NR>(a0, w0, acc);
            __builtin_amdgcn_sched_group_barrier(0x8, 5, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk<MR, NR>(arow, wp, (c + 2) & 31, a0, w0);
            mfma_chunk<MR, NR>(a1, w1, acc);
            __builtin_amdgcn_sched_group_barrier(0x8, 5, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (VAR == 12) {   // interleave pattern M1 V1 D2 M7
            load_chunk<MR, NR>(arow, wp, c + 1, a1, w1);
            mfma_chunk<MR, NR>(a0, w0, acc);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 7, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk<MR, NR>(arow, wp, (c + 2) & 31, a0, w0);
            mfma_chunk<MR, NR>(a1, w1, acc);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 7, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (VAR == 13) {   // interleave pattern M3 V1 M1 D2 M4
            load_chunk<MR, NR>(arow, wp, c + 1, a1, w1);
            mfma_chunk<MR, NR>(a0, w0, acc);
            __builtin_amdgcn_sched_group_barrier(0x8, 3, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk<MR, NR>(arow, wp, (c + 2) & 31, a0, w0);
            mfma_chunk<MR, NR>(a1, w1, acc);
            __builtin_amdgcn_sched_group_barrier(0x8, 3, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (VAR == 4) {   // only the weight loads
            w1[0] = wp[(c + 1) * 64];
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk<MR, NR>(a0, w0, acc);
            __builtin_amdgcn_sched_barrier(0);
            w0[0] = wp[((c + 2) & 31) * 64];
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk<MR, NR>(a0, w1, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__device__ __forceinline__ void gemm_deep(const float* Hw, const float4* Wp, int cb0, int lane, f32x16 (&acc)[2][1]) {
    constexpr int MR = 2, NR = 1;
    const float* arow = Hw + (lane & 31) * LDH + 4 * (lane >> 5);
    const float4* wp = Wp + (size_t)cb0 * (32 * 64) + lane;
    float4 a0[MR], a1[MR], w0[NR], w1[NR], w2[NR], w3[NR];
    w0[0] = wp[0]; w1[0] = wp[64]; w2[0] = wp[128];
#pragma unroll
    for (int i = 0; i < MR; ++i) a0[i] = *reinterpret_cast<const float4*>(arow + i * 32 * LDH);
#define ST(CW, WL, CA, AL, AU, WU)                                                        \
    WL[0] = wp[((CW) & 31) * 64];                                                         \
    _Pragma("unroll") for (int i = 0; i < MR; ++i) AL[i] = *reinterpret_cast<const float4*>(arow + i * 32 * LDH + 8 * ((CA) & 31)); \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    mfma_chunk<MR, NR>(AU, WU, acc);                                                      \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int c = 0; c < 32; c += 4) {
        ST(c + 3, w3, c + 1, a1, a0, w0)
        ST(c + 4, w0, c + 2, a0, a1, w1)
        ST(c + 5, w1, c + 3, a1, a0, w2)
        ST(c + 6, w2, c + 4, a0, a1, w3)
    }
#undef ST
}

#define IL_() __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           \
    __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           \
    __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_barrier(0);

__device__ __forceinline__ void gemm_u4(const float* Hw, const float4* Wp, int cb0, int lane, f32x16 (&acc)[2][1]) {
    constexpr int MR = 2, NR = 1;
    const float* ar = Hw + (lane & 31) * LDH + 4 * (lane >> 5);
    const float4* wp = Wp + (size_t)cb0 * (32 * 64) + lane;
    float4 a0[MR], a1[MR], w0[NR], w1[NR];
    load_chunk<MR, NR>(ar, wp, 0, a0, w0);
#pragma unroll 1
    for (int c = 0; c < 32; c += 4) {
        load_chunk<MR, NR>(ar, wp, c + 1, a1, w1); mfma_chunk<MR, NR>(a0, w0, acc); IL_()
        load_chunk<MR, NR>(ar, wp, c + 2, a0, w0); mfma_chunk<MR, NR>(a1, w1, acc); IL_()
        load_chunk<MR, NR>(ar, wp, c + 3, a1, w1); mfma_chunk<MR, NR>(a0, w0, acc); IL_()
        load_chunk<MR, NR>(ar, wp, (c + 4) & 31, a0, w0); mfma_chunk<MR, NR>(a1, w1, acc); IL_()
    }
}

// buffer loads for the weights: SGPR descriptor + per-lane byte offset, immediate chunk offsets
__device__ __forceinline__ void gemm_buf(const float* Hw, const float4* Wp, int cb0, int lane, f32x16 (&acc)[2][1]) {
    constexpr int MR = 2, NR = 1;
    const float* ar = Hw + (lane & 31) * LDH + 4 * (lane >> 5);
    const float4* base = Wp + (size_t)__builtin_amdgcn_readfirstlane(cb0) * (32 * 64);
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 32 * 64 * 16, 0x00020000);
    const int voff = lane * 16;
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto ldw = [&](int c) -> float4 { f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, c * 1024, 0)); return make_float4(v.x, v.y, v.z, v.w); };
    float4 a0[MR], a1[MR], w0[NR], w1[NR];
    w0[0] = ldw(0);
#pragma unroll
    for (int i = 0; i < MR; ++i) a0[i] = *reinterpret_cast<const float4*>(ar + i * 32 * LDH);
#pragma unroll 1
    for (int c = 0; c < 32; c += 2) {
        w1[0] = ldw(c + 1);
#pragma unroll
        for (int i = 0; i < MR; ++i) a1[i] = *reinterpret_cast<const float4*>(ar + i * 32 * LDH + 8 * (c + 1));
        mfma_chunk<MR, NR>(a0, w0, acc); IL_()
        w0[0] = ldw((c + 2) & 31);
#pragma unroll
        for (int i = 0; i < MR; ++i) a0[i] = *reinterpret_cast<const float4*>(ar + i * 32 * LDH + 8 * ((c + 2) & 31));
        mfma_chunk<MR, NR>(a1, w1, acc); IL_()
    }
}

template <int VAR>
__global__ __launch_bounds__(512) void k(const float4* W, float* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) float Hs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * LDH; i += 512) Hs[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    for (int r = 0; r < reps; ++r) {
        if (VAR == 14) gemm_u4(Hs, W + (size_t)(r % 3) * (8 * 32 * 64), wave, lane, acc);
        else if (VAR == 15) gemm_buf(Hs, W + (size_t)(r % 3) * (8 * 32 * 64), wave, lane, acc);
        else if (VAR == 5) gemm_deep(Hs, W + (size_t)(r % 3) * (8 * 32 * 64), wave, lane, acc);
        else gemm_var<VAR>(Hs, W + (size_t)(r % 3) * (8 * 32 * 64), wave, lane, acc);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][0][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <int VAR>
void run(const char* name, const float4* W, float* out, int wgs_per_cu) {
    const size_t lds = 64 * LDH * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int reps = 300;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(256 * wgs_per_cu), dim3(512), lds, 0, W, out, 3);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<VAR>, dim3(256 * wgs_per_cu), dim3(512), lds, 0, W, out, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)reps * 256 * 2 * wgs_per_cu;   // 256 MFMAs per wave per GEMM, 2 waves per SIMD per WG
    printf("%-34s WGs/CU %d : %.1f cycles per MFMA per SIMD, %.1f TFLOP/s\n", name, wgs_per_cu, ms * 1e-3 * 2.4e9 / mfma_per_simd,
           256.0 * 4 * mfma_per_simd * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
    float4* W; float* out;
    hipMalloc(&W, 3 * 8 * 32 * 64 * sizeof(float4)); hipMemset(W, 0, 3 * 8 * 32 * 64 * sizeof(float4));
    hipMalloc(&out, 1024 * 512 * 4);
    for (int g : {1, 2}) {
        run<0>("shipped (loads | setprio | mfma)", W, out, g);
        run<1>("no setprio", W, out, g);
        run<2>("mfma only", W, out, g);
        run<3>("LDS reads only", W, out, g);
        run<4>("weight loads only", W, out, g);
        run<5>("weights 3 chunks ahead", W, out, g);
        run<6>("mfma | both loads | mfma", W, out, g);
        run<7>("no sched barriers", W, out, g);
        run<8>("sched_group_barrier interleave", W, out, g);
        run<14>("interleave, 4 stages per iteration", W, out, g);
        run<15>("interleave, buffer_load weights", W, out, g);
        run<9>("M1 V1 M1 D1 M1 D1 M5", W, out, g);
        run<10>("M2 V1 M2 D1 M2 D1 M2", W, out, g);
        run<11>("M5 V1 M1 D1 M1 D1 M1", W, out, g);
        run<12>("M1 V1 D2 M7", W, out, g);
        run<13>("M3 V1 M1 D2 M4", W, out, g);

    }
    return 0;
}

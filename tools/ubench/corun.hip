// Microbenchmark: how much do the "other" instruction classes (VALU, LDS, global loads, cross-lane permutes) slow down
// when two waves per SIMD stream back-to-back v_mfma_f32_32x32x2_f32 next to them?  One 1024-thread workgroup per CU:
// waves 0-7 (2 per SIMD) are role A = MFMA stream, waves 8-15 role B = the probed class (or the other way round).
// Prints B's time per operation with A idle and with A streaming, at equal and at raised priority.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { B_VALU = 0, B_LDSW = 1, B_LDSR = 2, B_VMEM = 3, B_PERM = 4, B_MFMA16 = 5, B_N = 6 };
static const char* names[B_N] = {"VALU fma chain", "ds_write_b128", "ds_read_b128", "global_load_dwordx4 (L2)", "ds_bpermute", "dependent 16x16x4 MFMA"};

// s_nop k idles the wave for (k+1) x 4 clocks.  AMODE = 10 + idle clocks / 4 after EVERY MFMA: the wave does not present its next
// MFMA (which would sit in the VALU issue stage and block the other waves' VALU) until shortly before the pipe frees.
template <int AMODE>
__device__ __forceinline__ void spacer() {
    if constexpr (AMODE >= 10) {
        constexpr int q = AMODE - 10;            // quad-cycles
        if constexpr (q >= 16) asm volatile("s_nop 15");
        if constexpr (q >= 32) asm volatile("s_nop 15");
        if constexpr (q % 16 == 4) asm volatile("s_nop 3");
        if constexpr (q % 16 == 8) asm volatile("s_nop 7");
        if constexpr (q % 16 == 12) asm volatile("s_nop 11");
    }
}

template <int AMODE>
__global__ __launch_bounds__(1024) void k(int a_on, int a_iters, int b_kind, int b_ops, int prio, int a_first,
                                           const float4* __restrict__ src, float* out, long long* tim) {
    __shared__ float4 lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool roleA = a_first ? (wave < 8) : (wave >= 8);
    for (int i = threadIdx.x; i < 4096; i += 1024) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    float s = 0.f;
    if (roleA) {
        if (!a_on) return;
        const long long ta0 = wall_clock64();
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        for (int it = 0; it < a_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.f, 1.f, acc0, 0, 0, 0);
                spacer<AMODE>();
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.f, 1.f, acc1, 0, 0, 0);
                if (AMODE == 4) asm volatile("s_nop 0");          // one idle issue slot after every MFMA pair
                spacer<AMODE>();
            }
            if (AMODE == 1) __builtin_amdgcn_s_sleep(1);            // ~64 cycles every 16 MFMAs
            if (AMODE == 2) asm volatile("s_nop 15");
            if (AMODE == 3) { __builtin_amdgcn_s_setprio(0); }
            if (AMODE == 5) __builtin_amdgcn_sched_barrier(0), __builtin_amdgcn_s_sleep(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
        out[blockIdx.x * 1024 + threadIdx.x] = s;
        if (lane == 0) tim[blockIdx.x * 16 + wave] = wall_clock64() - ta0;
        return;
    }
    if (prio) __builtin_amdgcn_s_setprio(3);
    // let role A get going
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(10);
    const long long t0 = wall_clock64();
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b_kind == B_VALU) {
        float x = (float)lane;
#pragma unroll 8
        for (int i = 0; i < b_ops; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
        s = x;
    } else if (b_kind == B_LDSW) {
        for (int i = 0; i < b_ops; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) lds[((wave & 7) * 512 + u * 64 + lane) & 4095] = make_float4((float)i, 0.f, 0.f, (float)u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if (b_kind == B_LDSR) {
        for (int i = 0; i < b_ops; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float4 t; { const float4* p = &lds[((wave & 7) * 512 + u * 64 + lane + (i & 1)) & 4095]; t = *p; asm volatile("" : "+v"(t.x), "+v"(t.w)); }
                v.x += t.x; v.y += t.w;
            }
        }
        s = v.x + v.y;
    } else if (b_kind == B_VMEM) {
        for (int i = 0; i < b_ops; i += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(((i + u) * 16 + wave) & 16383) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v.x += t[u].x; v.y += t[u].w; }
        }
        s = v.x + v.y;
    } else if (b_kind == B_PERM) {
        float x = (float)lane;
        for (int i = 0; i < b_ops; ++i) x = fminf(x + 1.f, __shfl_xor(x, 1 + (i & 7)));
        s = x;
    } else {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int i = 0; i < b_ops; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, 1.f, acc, 0, 0, 0);
        s = acc[0] + acc[3];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t1 = wall_clock64();
    if (lane == 0) tim[blockIdx.x * 16 + wave] = t1 - t0;
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int AMODE>
void run_mode(const char* label, float* out, long long* tim, float4* src, int ops, int a_iters) {
    printf("A mode: %s  (A alone would take %.3f ms; B ops %d)\n", label, a_iters * 16.0 * 2 * 64 / 2.4e6, ops);
    for (int kind = 0; kind < B_N; ++kind) {
        double resB[3], resA[3];
        for (int cfg = 0; cfg < 3; ++cfg) {
            const int a_on = cfg > 0, prio = cfg == 2;
            (void)hipMemset(tim, 0, 256 * 16 * 8);
            hipLaunchKernelGGL(k<AMODE>, dim3(256), dim3(1024), 0, 0, a_on, a_iters, kind, ops, prio, 1, src, out, tim);
            (void)hipDeviceSynchronize();
            std::vector<long long> h(256 * 16);
            (void)hipMemcpy(h.data(), tim, 256 * 16 * 8, hipMemcpyDeviceToHost);
            double sa = 0, sb = 0; int na = 0, nb = 0;
            for (int i = 0; i < 256 * 16; ++i) {
                if (h[i] <= 0) continue;
                if ((i & 15) < 8) { sa += (double)h[i]; ++na; } else { sb += (double)h[i]; ++nb; }
            }
            resB[cfg] = nb ? sb / nb * 10.0 / 1e6 : 0.0;                       // B total ms
            resA[cfg] = na ? sa / na * 10.0 * 2.4 / (a_iters * 16.0 * 2) : 0.0;   // cycles per MFMA per SIMD (2 A waves per SIMD)
        }
        printf("  %-28s B total ms: alone %8.3f  with A %8.3f  with A, prio 3 %8.3f | A cycles/MFMA %6.1f %6.1f\n", names[kind], resB[0], resB[1],
               resB[2], resA[1], resA[2]);
    }
}

int main(int argc, char** argv) {
    const int ops = argc > 1 ? atoi(argv[1]) : 4096, a_iters = argc > 2 ? atoi(argv[2]) : 4000;
    float* out; long long* tim; float4* src;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&tim, 256 * 16 * 8); (void)hipMalloc(&src, (size_t)16384 * 64 * 16);
    (void)hipMemset(src, 0, (size_t)16384 * 64 * 16);
    run_mode<0>("back-to-back MFMA", out, tim, src, ops, a_iters);
    run_mode<1>("s_sleep 1 every 16 MFMAs", out, tim, src, ops, a_iters);
    run_mode<10 + 8>("32 idle clocks after every MFMA", out, tim, src, ops, a_iters);
    run_mode<10 + 12>("48 idle clocks after every MFMA", out, tim, src, ops, a_iters);
    run_mode<10 + 16>("64 idle clocks after every MFMA", out, tim, src, ops, a_iters);
    run_mode<10 + 20>("80 idle clocks after every MFMA", out, tim, src, ops, a_iters);
    run_mode<10 + 24>("96 idle clocks after every MFMA", out, tim, src, ops, a_iters);
    run_mode<10 + 28>("112 idle clocks after every MFMA", out, tim, src, ops, a_iters);
    return 0;
}

// One 16-row tile per CU (k_tail_sel, k_step_small): each wave streams its 32 KB share of a 256 KB weight layer out of L2,
// 1 KB per load, 2 loads per k-chunk, and spends 8 v_mfma_f32_16x16x4_f32 on a chunk.  What does the stream cost as a
// function of how many chunks a wave keeps in flight, with and without the MFMAs, and does the order of the loads inside
// the pack matter (L2 channel hot spot: every CU reads the same addresses at the same time)?
//   layout 0: [colblk16][kchunk][lane]   wave w reads blocks 2w, 2w+1: the 8 waves' loads of chunk c are 32 KB apart
//   layout 1: [kchunk][colblk16][lane]   the 8 waves of a workgroup read one contiguous 16 KB per chunk
//   DEPTH = chunks in flight per wave (1 = gemm16 as shipped).  Loads wrap around the pack, so every variant issues
//   (layers * 16 + DEPTH) chunks; the time is reported per layer of 16 chunks.
//   OOB = 1: the wrap-around loads are out-of-range buffer loads (VGPR offset 0xffffffff) instead.
// Build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o tools/ubench/l2_hotspot tools/ubench/l2_hotspot.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LAYER = 16 * 16 * 64;   // float4 per layer

template <int LAYOUT, int DEPTH, int MF, int OOB>
__global__ __launch_bounds__(512) void k(const f4* __restrict__ W, int nl, float* out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4*>(W), 0, nl * LAYER * 16, 0x00020000);
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    f4 ring[DEPTH][2];
    const int total = nl * 16;
    auto load = [&](int s, int j) {
        const bool wrap = s >= total;
        if (wrap) s -= total;
        const int l = s >> 4, c = s & 15;
        const int b = 2 * wave + j;
        const int off = (LAYOUT == 0 ? (b * 16 + c) * 64 : (c * 16 + b) * 64) * 16 + l * LAYER * 16;
        const int voff = (OOB && wrap) ? -1 : lane * 16;
        return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, off, 0));
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
        ring[s][0] = load(s, 0);
        ring[s][1] = load(s, 1);
        __builtin_amdgcn_sched_barrier(0);   // issue order = consumption order, or the first wait of the loop is vmcnt(0)
    }
    const float a = (float)lane;
    for (int s0 = 0; s0 < total; s0 += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            if (MF) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[i][0].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[i][1].x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[i][0].y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[i][1].y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[i][0].z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[i][1].z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[i][0].w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ring[i][1].w, acc[1], 0, 0, 0);
            } else {
                acc[0] += ring[i][0];
                acc[1] += ring[i][1];
            }
            ring[i][0] = load(s0 + i + DEPTH, 0);
            ring[i][1] = load(s0 + i + DEPTH, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const f4 r = acc[0] + acc[1];
    if (r.x + r.y + r.z + r.w == 12345.f) out[0] = r.x;
}

template <int LAYOUT, int DEPTH, int MF, int OOB>
float run(const f4* W, int nl, float* out, int wgs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LAYOUT, DEPTH, MF, OOB>), dim3(wgs), dim3(512), 0, 0, W, nl, out);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<LAYOUT, DEPTH, MF, OOB>), dim3(wgs), dim3(512), 0, 0, W, nl, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}

template <int LAYOUT, int MF, int OOB>
void row(const f4* W, float* out, int wgs) {
    // two pack sizes: the difference is the cost of 12 more layers, free of the launch overhead
    auto per_layer = [&](auto fn) { return (fn(15) - fn(3)) / 12.f; };
    const float d1 = per_layer([&](int nl) { return run<LAYOUT, 1, MF, OOB>(W, nl, out, wgs); });
    const float d2 = per_layer([&](int nl) { return run<LAYOUT, 2, MF, OOB>(W, nl, out, wgs); });
    const float d4 = per_layer([&](int nl) { return run<LAYOUT, 4, MF, OOB>(W, nl, out, wgs); });
    const float d8 = per_layer([&](int nl) { return run<LAYOUT, 8, MF, OOB>(W, nl, out, wgs); });
    const float d16 = per_layer([&](int nl) { return run<LAYOUT, 16, MF, OOB>(W, nl, out, wgs); });
    printf("wgs=%3d layout %d mfma %d oob %d: us per 256 KB layer at depth 1/2/4/8/16 = %.2f %.2f %.2f %.2f %.2f   (3-layer launch, depth 1 / 16: %.2f / %.2f us)\n", wgs, LAYOUT, MF,
           OOB, d1, d2, d4, d8, d16, run<LAYOUT, 1, MF, OOB>(W, 3, out, wgs), run<LAYOUT, 16, MF, OOB>(W, 3, out, wgs));
}

// ---- cold L2: every CU of an XCD asks for the same missing lines at the same time ------------------------------------
__global__ void k_evict(const f4* __restrict__ E, size_t n, float* out) {
    f4 s = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += E[i];
    if (s.x == 12345.f) out[1] = s.x;
}
// each workgroup touches its own 1/32 of the pack (workgroups go round the 8 XCDs, so blockIdx / 8 numbers the CUs of an XCD):
// a line is asked for once per XCD instead of 32 times
__global__ __launch_bounds__(512) void k_touch(const float* __restrict__ W, int bytes, float* out) {
    const int slice = bytes / 32, cu = (blockIdx.x >> 3) & 31;
    float s = 0.f;
    for (int o = threadIdx.x * 128; o < slice; o += 512 * 128) s += W[(cu * slice + o) >> 2];
    if (s == 12345.f) out[2] = s;
}
template <int DEPTH>
void cold_row(const f4* W, const f4* E, size_t en, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int nl = 3;
    auto timed = [&](int mode) {   // 0 warm, 1 cold, 2 cold + touch kernel in front (timed together), 3 cold + touch kernel (not timed)
        float best = 1e9f, sum = 0.f;
        const int reps = 10;
        for (int i = 0; i < reps; ++i) {
            if (mode) hipLaunchKernelGGL(k_evict, dim3(2048), dim3(256), 0, 0, E, en, out);
            if (mode == 3) hipLaunchKernelGGL(k_touch, dim3(256), dim3(512), 0, 0, (const float*)W, nl * LAYER * 16, out);
            hipEventRecord(e0);
            if (mode == 2) hipLaunchKernelGGL(k_touch, dim3(256), dim3(512), 0, 0, (const float*)W, nl * LAYER * 16, out);
            hipLaunchKernelGGL((k<0, DEPTH, 1, 0>), dim3(256), dim3(512), 0, 0, W, nl, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
            sum += ms;
        }
        return sum * 1000.f / reps;
    };
    timed(0);
    const float w = timed(0), c = timed(1), ct = timed(2), cp = timed(3);
    printf("256 wgs, 3 layers with MFMAs, depth %2d: warm %.2f us, cold L2 %.2f, cold + touch kernel (timed) %.2f, cold + touch kernel (before the clock) %.2f\n", DEPTH, w, c, ct, cp);
}

int main() {
    const int nl = 15;
    f4* W;
    float* out;
    hipMalloc(&W, (size_t)nl * LAYER * sizeof(f4));
    hipMalloc(&out, 4);
    std::vector<float> h((size_t)nl * LAYER * 4, 1.f);
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    {
        f4* E;
        const size_t en = (size_t)512 << 20 >> 4;
        hipMalloc(&E, en * sizeof(f4));
        hipMemset(E, 0, en * sizeof(f4));
        cold_row<1>(W, E, en, out);
        cold_row<2>(W, E, en, out);
        cold_row<16>(W, E, en, out);
        hipFree(E);
    }
    for (int wgs : {256}) {
        row<0, 0, 0>(W, out, wgs);
        row<1, 0, 0>(W, out, wgs);
        row<0, 1, 0>(W, out, wgs);
        row<1, 1, 0>(W, out, wgs);
        row<0, 1, 1>(W, out, wgs);
    }
    return 0;
}

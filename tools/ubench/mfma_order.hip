// In which order do v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 add their K products to the accumulator?
// One output element per test: acc0 + sum_k a_k b_k with values chosen so that every association gives a different fp32
// result; compared with sequential fmaf chains and with pairwise / tree orders on the host.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k32(const float* a, const float* b, float c0, float* out) {   // A[32][2], B[2][32]; lane l: a = A[l&31][l>>5], b = B[l>>5][l&31]
    const int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c0;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(l & 31) * 2 + (l >> 5)], b[(l >> 5) * 32 + (l & 31)], acc, 0, 0, 0);
    if (l == 0) out[0] = acc[0];   // element (row 0, col 0)
}
__global__ void k16(const float* a, const float* b, float c0, float* out) {   // A[16][4], B[4][16]; lane l: a = A[l&15][l>>4], b = B[l>>4][l&15]
    const int l = threadIdx.x;
    f32x4 acc = {c0, c0, c0, c0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(l & 15) * 4 + (l >> 4)], b[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    if (l == 0) out[0] = acc[0];
}

int main() {
    float *da, *db, *dout;
    hipMalloc(&da, 64 * 4); hipMalloc(&db, 64 * 4); hipMalloc(&dout, 4);
    unsigned st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((int)(st >> 8) - (1 << 23)) * (1.0f / (1 << 20)); };
    int n32[4] = {0, 0, 0, 0}, n16[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int T = 2000;
    for (int t = 0; t < T; ++t) {
        float a[64], b[64];
        for (int i = 0; i < 64; ++i) { a[i] = rnd(); b[i] = rnd(); }
        const float c0 = rnd() * 3.f;
        float o32, o16;
        hipMemcpy(da, a, 256, hipMemcpyHostToDevice); hipMemcpy(db, b, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, da, db, c0, dout); hipMemcpy(&o32, dout, 4, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, da, db, c0, dout); hipMemcpy(&o16, dout, 4, hipMemcpyDeviceToHost);
        // 32x32x2, element (0,0): products p0 = A[0][0] B[0][0], p1 = A[0][1] B[1][0]
        {
            const float a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[32];
            const float seq01 = fmaf(a1, b1, fmaf(a0, b0, c0)), seq10 = fmaf(a0, b0, fmaf(a1, b1, c0));
            const float pair = (float)((double)a0 * b0 + (double)a1 * b1) + c0;                     // products summed exactly, then + c
            const float exact = (float)((double)c0 + (double)a0 * b0 + (double)a1 * b1);             // single rounding
            n32[0] += o32 == seq01; n32[1] += o32 == seq10; n32[2] += o32 == pair; n32[3] += o32 == exact;
        }
        // 16x16x4, element (0,0): p_g = A[0][g] B[g][0]
        {
            const float av[4] = {a[0], a[1], a[2], a[3]}, bv[4] = {b[0], b[16], b[32], b[48]};
            float s = c0; for (int g = 0; g < 4; ++g) s = fmaf(av[g], bv[g], s);
            float r = c0; for (int g = 3; g >= 0; --g) r = fmaf(av[g], bv[g], r);
            double e = c0; for (int g = 0; g < 4; ++g) e += (double)av[g] * bv[g];
            const float tree = (float)(((double)av[0] * bv[0] + (double)av[1] * bv[1]) + ((double)av[2] * bv[2] + (double)av[3] * bv[3])) + c0;
            // two sequential 2-product steps, each like 32x32x2 if that one is "pair then add"
            const float two = (float)((double)(float)((double)c0 + (double)av[0] * bv[0] + (double)av[1] * bv[1]) + (double)av[2] * bv[2] + (double)av[3] * bv[3]);
            n16[0] += o16 == s; n16[1] += o16 == r; n16[2] += o16 == (float)e; n16[3] += o16 == tree; n16[4] += o16 == two;
        }
    }
    printf("32x32x2 : of %d random cases equal to  fma chain k=0,1: %d   k=1,0: %d   (p0+p1 exact)+c: %d   single rounding: %d\n", T, n32[0], n32[1], n32[2], n32[3]);
    printf("16x16x4 : of %d random cases equal to  fma chain k=0..3: %d   k=3..0: %d   single rounding: %d   tree+c: %d   two exact pairs: %d\n", T, n16[0], n16[1], n16[2], n16[3], n16[4]);
    return 0;
}

// sin / cos of the positional encoding [x, sin x, cos x] in the REFERENCE's arithmetic.
//
// network_macros_mod.py:139-140 calls torch.sin / torch.cos on a contiguous fp32 CPU tensor; torch 2.10 (the version pinned in the
// build container) evaluates them with SLEEF 3.x's 1.0-ULP kernels Sleef_sinf16_u10 / Sleef_cosf16_u10 (xsinf_u1 / xcosf_u1 of
// sleefsimdsp.c, AVX-512 build with FMA), not with a correctly rounded sin: about 15 % of its values differ from glibc's, numpy's
// and ocml's by one ulp, and an input feature that is one ulp off moves every unit of layer 1.  This is a restatement of the published
// algorithm for |x| < 125 (TRIGRANGEMAX2f; the Payne-Hanek branch above it is not restated -- joint angles and obstacle coordinates
// in metres are nowhere near -- and falls back to the device library's sinf / cosf): Cody-Waite reduction by pi in three parts with
// the remainder kept as a double-float, a degree-3 polynomial in s^2 evaluated in double-float arithmetic, every operation written
// out so that no compiler contraction can change a bit.  Checked bit for bit against torch.sin / torch.cos on the CPU
// (tests/test_trig_cpu.py: the same header compiled for the host by g++) and on the device against the host build
// (tests/test_gpu_trig.py).
//
// The double-float helpers are SLEEF's df.h in its FMA form (the form the AVX-512 build uses).
#pragma once

#if defined(__HIPCC__) || defined(__CUDACC__)
#define OMDS_TRIG_FN __host__ __device__ __forceinline__
#else
#define OMDS_TRIG_FN static inline
#endif

namespace omds_trig {

struct f2 { float x, y; };

// plain IEEE operations, one rounding each: every fused operation is an explicit __builtin_fmaf and every function body switches
// contraction off (the library is built with -ffp-contract=on), so a + b * c is never fused behind our back
#if defined(__clang__)
#define OMDS_NOCONTRACT _Pragma("clang fp contract(off)")
#else
#define OMDS_NOCONTRACT   /* g++ host builds of this header pass -ffp-contract=off on the command line (tests/test_trig_cpu.py) */
#endif

OMDS_TRIG_FN float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// dfadd2_vf2_vf_vf: x + y as a double-float, no assumption on the magnitudes
OMDS_TRIG_FN f2 add2_ff(float x, float y) {
    OMDS_NOCONTRACT
    f2 r;
    r.x = x + y;
    const float v = r.x - x;
    r.y = (x - (r.x - v)) + (y - v);
    return r;
}
// dfadd2_vf2_vf2_vf
OMDS_TRIG_FN f2 add2_f2f(f2 x, float y) {
    OMDS_NOCONTRACT
    f2 r;
    r.x = x.x + y;
    const float v = r.x - x.x;
    const float w = (x.x - (r.x - v)) + (y - v);
    r.y = x.y + w;
    return r;
}
// dfadd_vf2_vf2_vf: |x| >= |y|
OMDS_TRIG_FN f2 add_f2f(f2 x, float y) {
    OMDS_NOCONTRACT
    f2 r;
    r.x = x.x + y;
    r.y = x.y + ((x.x - r.x) + y);
    return r;
}
// dfadd_vf2_vf_vf: |x| >= |y|
OMDS_TRIG_FN f2 add_ff(float x, float y) {
    OMDS_NOCONTRACT
    f2 r;
    r.x = x + y;
    r.y = (x - r.x) + y;
    return r;
}
// dfadd_vf2_vf_vf2: |x| >= |y|
OMDS_TRIG_FN f2 add_ff2(float x, f2 y) {
    OMDS_NOCONTRACT
    f2 r;
    r.x = x + y.x;
    r.y = ((x - r.x) + y.x) + y.y;
    return r;
}
// dfsqu_vf2_vf2
OMDS_TRIG_FN f2 squ(f2 x) {
    OMDS_NOCONTRACT
    f2 r;
    r.x = x.x * x.x;
    r.y = fma_(x.x + x.x, x.y, fma_(x.x, x.x, -r.x));
    return r;
}
// dfmul_vf2_vf2_vf2
OMDS_TRIG_FN f2 mul(f2 x, f2 y) {
    OMDS_NOCONTRACT
    f2 r;
    r.x = x.x * y.x;
    r.y = fma_(x.x, y.y, fma_(x.y, y.x, fma_(x.x, y.x, -r.x)));
    return r;
}
// dfmul_vf_vf2_vf2: the product rounded to one float
OMDS_TRIG_FN float mul_to_f(f2 x, f2 y) {
    OMDS_NOCONTRACT
    const float p = x.x * y.y;
    return fma_(x.x, y.x, fma_(x.y, y.x, p));
}

constexpr float PI_A2 = 3.1414794921875f, PI_B2 = 0.00011315941810607910156f, PI_C2 = 1.9841872589410058936e-09f;
constexpr float M_1_PI_F = 0.318309886183790671537767526745028724f;
constexpr float RANGE_MAX2 = 125.0f;

// the kernel both functions share: sin of the reduced argument t (a double-float, |t| <= pi/2) = t (1 + s (c3 + s poly(s))), s = t^2
OMDS_TRIG_FN float sin_reduced(f2 t) {
    OMDS_NOCONTRACT
    const f2 s = squ(t);
    float u = 2.6083159809786593541503e-06f;
    u = fma_(u, s.x, -0.0001981069071916863322258f);
    u = fma_(u, s.x, 0.00833307858556509017944336f);
    const f2 x = add_ff2(1.0f, mul(add_ff(-0.166666597127914428710938f, u * s.x), s));
    return mul_to_f(t, x);
}

OMDS_TRIG_FN bool in_range(float d) { return __builtin_fabsf(d) < RANGE_MAX2; }

// xsinf_u1, |d| < 125
OMDS_TRIG_FN float sin_u10(float d) {
    OMDS_NOCONTRACT
    const float u = __builtin_rintf(d * M_1_PI_F);
    const int q = (int)u;
    const float v = fma_(u, -PI_A2, d);
    f2 s = add2_ff(v, u * -PI_B2);
    s = add_f2f(s, u * -PI_C2);
    float r = sin_reduced(s);
    if (q & 1) r = -r;
    return (d == 0.0f && __builtin_signbit(d)) ? d : r;   // sin(-0) = -0
}

// xcosf_u1, |d| < 125
OMDS_TRIG_FN float cos_u10(float d) {
    OMDS_NOCONTRACT
    const float dq = fma_(__builtin_rintf(fma_(d, M_1_PI_F, -0.5f)), 2.0f, 1.0f);
    const int q = (int)dq;
    f2 s = add2_ff(d, dq * (-PI_A2 * 0.5f));
    s = add2_f2f(s, dq * (-PI_B2 * 0.5f));
    s = add2_f2f(s, dq * (-PI_C2 * 0.5f));
    float r = sin_reduced(s);
    if ((q & 2) == 0) r = -r;
    return r;
}

}   // namespace omds_trig

// the encoding's sin / cos: the reference's arithmetic where it is restated, the platform's elsewhere (never reached by this path)
OMDS_TRIG_FN float omds_sinf(float x) { return omds_trig::in_range(x) ? omds_trig::sin_u10(x) : __builtin_sinf(x); }
OMDS_TRIG_FN float omds_cosf(float x) { return omds_trig::in_range(x) ? omds_trig::cos_u10(x) : __builtin_cosf(x); }

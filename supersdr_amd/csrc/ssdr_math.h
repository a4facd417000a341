// ssdr_math.h -- deterministic fp32 device math for the SuperSDR hot path (gfx950).
//
// No libm transcendentals and no implicit contraction (-ffp-contract=off): every
// rounding step is explicit, so results are reproducible op-for-op on any IEEE-754
// machine.  v_fma_f32 / v_rndne_f32 / IEEE sqrt and integer bit operations are the only primitives.
#pragma once
#include <hip/hip_runtime.h>
#include "ssdr_consts.h"

#define SSDR_DEV __device__ __forceinline__
// streaming accesses (A/B switches: -DSSDR_PLAIN_STORES / -DSSDR_PLAIN_LOADS)
#ifdef SSDR_PLAIN_STORES
#define SSDR_NT_STORE(v, p) (*(p) = (v))
#else
#define SSDR_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif
#ifdef SSDR_PLAIN_LOADS
#define SSDR_NT_LOAD(p) (*(p))
#else
#define SSDR_NT_LOAD(p) __builtin_nontemporal_load(p)
#endif

// sin/cos of 2*pi*(phase>>12)/2^20: 20-bit phase truncation (DDS practice), exact
// integer quadrant reduction, minimax polynomials on [-pi/4, pi/4].
SSDR_DEV void ssdr_sincos20(uint32_t phase, float &c_out, float &s_out)
{
    const float S1 = -1.6666654611e-1f, S2 = 8.3321608736e-3f, S3 = -1.9515295891e-4f;
    const float C1 = 4.166664568298827e-2f, C2 = -1.388731625493765e-3f, C3 = 2.443315711809948e-5f;
    uint32_t p20 = phase >> 12;
    uint32_t k = (p20 + (1u << 17)) >> 18;
    int32_t ri = (int32_t)p20 - (int32_t)(k << 18);
    float th = (float)ri * SSDR_C_2PI_20;
    float t2 = th * th;
    float u = fmaf(t2, S3, S2);
    u = fmaf(t2, u, S1);
    float s = fmaf(th * t2, u, th);
    float v = fmaf(t2, C3, C2);
    v = fmaf(t2, v, C1);
    float c = fmaf(t2 * t2, v, fmaf(-0.5f, t2, 1.0f));
    // rotate by k quadrants: branch-free selects
    float cc = (k & 1u) ? s : c;
    float ss = (k & 1u) ? c : s;
    c_out = ((k + 1u) & 2u) ? -cc : cc;       // k=1,2 -> negative cos side
    s_out = (k & 2u) ? -ss : ss;              // k=2,3 -> negative sin side
}

// cos/sin of 2 pi phase / 2^32 at the full 32 bits of the phase: the 20-bit evaluation above plus the first-order term
// for the 12 bits below it (|eps| < 6e-6, eps^2/2 < 2e-11 -- under fp32 rounding).
SSDR_DEV void ssdr_phasor32(uint32_t phase, float &c, float &s)
{
    float c20, s20;
    ssdr_sincos20(phase, c20, s20);
    const float eps = (float)(phase & 0xFFFu) * SSDR_C_2PI_32;
    c = fmaf(-s20, eps, c20);
    s = fmaf(c20, eps, s20);
}

// 1/d for d in [1, 2.42]: cubic seed (|rel err| < 5.5e-3) + two Newton steps (-> 1e-9, i.e. fp32 rounding is what is left).
// Seven full-rate ops; an IEEE divide is ~15 instructions, several of them quarter-rate.  The twin states the same ops.
SSDR_DEV float ssdr_rcp_1to2p42(float d)
{
    float r = fmaf(fmaf(fmaf(-0.1340303272008896f, d, 0.9271132946014404f), d, -2.337818145751953f), d, 2.539294958114624f);
    float e = fmaf(-d, r, 1.0f);
    r = fmaf(r, e, r);
    e = fmaf(-d, r, 1.0f);
    r = fmaf(r, e, r);
    return r;
}

// log2(x) for normal x > 0: exponent split + atanh series in s = (m-1)/(m+1).
SSDR_DEV float ssdr_log2p(float x)
{
    const float L1 = 0.33333333333f, L2 = 0.2f, L3 = 0.14285714286f, L4 = 0.11111111111f;
    uint32_t I = __float_as_uint(x);
    int32_t e = (int32_t)(I >> 23) - 127;
    float m = __uint_as_float((I & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) * ssdr_rcp_1to2p42(m + 1.0f);      // m + 1 in [1.707, 2.414]
    float z = s * s;
    float t = fmaf(z, L4, L3);
    t = fmaf(z, t, L2);
    t = fmaf(z, t, L1);
    t = z * t;
    float sk = s * SSDR_K2_LN2;
    return (float)e + fmaf(sk, t, sk);
}

// 2^y, |y| <= 126
SSDR_DEV float ssdr_exp2p(float y)
{
    const float E1 = 0.69314718056f, E2 = 0.24022650696f, E3 = 0.055504108665f,
                E4 = 0.0096181291076f, E5 = 0.0013333558146f, E6 = 0.00015403530393f,
                E7 = 0.000015252733805f;
    y = __builtin_amdgcn_fmed3f(y, -126.0f, 126.0f);       // clamp in one instruction (same value as min(max()))
    float n = rintf(y);
    float f = y - n;
    float r = fmaf(E7, f, E6);
    r = fmaf(r, f, E5);
    r = fmaf(r, f, E4);
    r = fmaf(r, f, E3);
    r = fmaf(r, f, E2);
    r = fmaf(r, f, E1);
    r = fmaf(r, f, 1.0f);
    return __uint_as_float(__float_as_uint(r) + ((uint32_t)(int32_t)n << 23));
}

// atan2(y, x).  One division-free quotient t = min/max in [0, 1] (exact power-of-two scaling of the larger magnitude into
// [1, 2), then the seven-FMA reciprocal above), atan(t) = t + t^3 Q(t^2) with a degree-6 minimax Q on the whole of [0, 1]
// (|error| < 1e-7, no second range reduction), and the octant put back with sign-bit arithmetic:
//     |y| > |x|: pi/2 - r     x < 0: pi - (.)     y < 0: -(.)      ==  copysign(base +- r, y)
// Signed zeros carry no phase: -0 counts as +0 (atan2(0, 0) = 0, atan2(-0, x < 0) = +pi).
SSDR_DEV float ssdr_atan2p(float y, float x)
{
    const float Q0 = -3.3331659436e-01f, Q1 = 1.9962704182e-01f, Q2 = -1.3976582885e-01f, Q3 = 9.7942389548e-02f,
                Q4 = -5.7773657143e-02f, Q5 = 2.3040184751e-02f, Q6 = -4.3554198928e-03f;
    const float PI_2 = 1.5707963267948966f, PI_1 = 3.14159265358979323f;
    x = x + 0.0f;                                 // -0 -> +0
    y = y + 0.0f;
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float mxc = fmaxf(mx, 1e-30f);         // no 0/0: both zero gives t = 0, r = 0, base 0
    const float sc = __uint_as_float(0x7F000000u - (__float_as_uint(mxc) & 0x7F800000u));
    const float t = (mn * sc) * ssdr_rcp_1to2p42(mxc * sc);
    const float z = t * t;
    float q = fmaf(Q6, z, Q5);
    q = fmaf(q, z, Q4);
    q = fmaf(q, z, Q3);
    q = fmaf(q, z, Q2);
    q = fmaf(q, z, Q1);
    q = fmaf(q, z, Q0);
    const float r = fmaf(t * z, q, t);            // atan(t) in [0, pi/4]
    const float d = ax - ay;                      // sign bit set <=> |y| > |x|
    const uint32_t flip = (__float_as_uint(d) ^ __float_as_uint(x)) & 0x80000000u;
    const float rs = __uint_as_float(__float_as_uint(r) ^ flip);
    float base = (x < 0.0f) ? PI_1 : 0.0f;
    base = (d < 0.0f) ? PI_2 : base;
    const float a = rs + base;                    // >= 0
    return __uint_as_float(__float_as_uint(a) | (__float_as_uint(y) & 0x80000000u));
}

// Correctly rounded sqrt for 0 <= p < 2^62 (what |z|^2 can be): the hardware estimate (v_sqrt_f32,
// <= 1 ulp) is settled by the sign of two exact residuals, fma(-s', s, p) for the neighbours
// s' = s -/+ 1 ulp -- the compiler's own IEEE scheme, but instead of its compare/select denormal
// pre-scaling the argument is always scaled by 2^64 (exact) so the residuals cannot underflow.
// Identical to sqrtf() bit for bit (checked exhaustively on the GPU by ssdr_selftest_sqrt).
SSDR_DEV float ssdr_sqrt_rn(float p)
{
    const float q = p * 0x1p64f;
    const float s = __builtin_amdgcn_sqrtf(q);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = fmaf(-sd, s, q), ru = fmaf(-su, s, q);
    float r = (rd <= 0.0f) ? sd : s;            // s == 0: sd is a NaN pattern, rd NaN, compare false -> stays 0
    r = (ru > 0.0f) ? su : r;
    return r * 0x1p-32f;
}

// The same for p == 0 or 1 <= p < 2^33 (an integer power I*I + Q*Q rounded to float), without the neighbours: there one
// Newton step from the reciprocal square root,  s = p y,  s' = fma(fma(-s, s, p), y/2, s),  already IS the correctly
// rounded root (checked over every float of [1, 2^33) by tools/ubench/sqrt_variants.hip and by ssdr_selftest_sqrt; it
// fails only far below 1, where the residual underflows).  p == 0: the clamp bit turns v_rsq's +inf into 1.0 (and leaves
// every y <= 1, i.e. every p >= 1, alone), so s = 0, residual 0, result +0.  Five instructions instead of eight.
SSDR_DEV float ssdr_sqrt_rn_int(float p)
{
    float y;
    asm("v_rsq_f32_e64 %0, %1 clamp" : "=v"(y) : "v"(p));
    const float s = p * y, h = 0.5f * y;
    return fmaf(fmaf(-s, s, p), h, s);
}

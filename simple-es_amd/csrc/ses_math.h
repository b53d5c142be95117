// ses_math.h -- deterministic fp32 elementary functions for the gfx950 kernels.
//
// Every function is an explicit tree of IEEE-754 correctly rounded operations (add, mul, fma,
// div, sqrt, rndne, cvt, integer ops).  No v_exp/v_log/v_sin/v_rcp approximations and no
// compiler contraction (build with -ffp-contract=off): the discrete-action rollouts are chaotic
// in the last ulp, and per-offspring returns must equal the CPU restatement exactly.
// hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt keeps '/' and sqrtf IEEE-exact
// (v_div_scale/v_div_fmas/v_div_fixup sequence).
//
// Coefficients: single-precision Cephes minimax sets (public domain).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SES_DEV __device__ __forceinline__
#else
#define SES_DEV static inline  // host compile of the device functions (tests/hostcheck only)
#endif

#if defined(__HIPCC__)
#define SES_TANH_TABLE_QUAL static __device__ const __attribute__((aligned(16)))
#endif
#include "ses_tanh_table.h"

namespace ses {

SES_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
SES_DEV uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
SES_DEV float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
// NaN-absorbing clamp helpers; same value as the oracle's ternaries for every input
SES_DEV float min_(float a, float b) { return __builtin_fminf(a, b); }
SES_DEV float max_(float a, float b) { return __builtin_fmaxf(a, b); }

// e^x, x clamped to [-86, 88]
SES_DEV float exp_(float x)
{
    x = min_(max_(x, -86.0f), 88.0f);
    const float k = __builtin_rintf(x * 0x1.715476p+0f);
    float r = fma_(k, -0.693359375f, x);
    r = fma_(k, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float e = fma_(p, r2, r);
    e = e + 1.0f;
    const int32_t ki = (int32_t)k;
    return u2f(f2u(e) + ((uint32_t)ki << 23));
}

// 16-byte table entry: c0 + u*(c1 + u*(c2 + u*c3)), u = fract(|x|*32), on [i/32, (i+1)/32)
struct alignas(16) TanhEntry {
    float c0, c1, c2, c3;
};

// tanh from the piecewise-cubic table (tools/gen_tanh_table.py); `tab` lives in LDS inside kernels.
// 9 plain VALU instructions + one ds_read_b128, no division, no transcendental unit.
// tanh split in three so that callers can batch the table reads of several independent evaluations
// (issue every ds_read_b128 first, evaluate the cubics afterwards: the LDS latency is paid once, not per value).
SES_DEV int32_t tanh_index(float x, float &u)
{
    const float t = min_(__builtin_fabsf(x), SES_TANH_XMAX) * SES_TANH_H_INV;  // exact scaling by 32
#if defined(__HIPCC__)
    u = __builtin_amdgcn_fractf(t);                                            // v_fract_f32, exact
#else
    u = t - __builtin_floorf(t);
#endif
    return (int32_t)t;
}

// same index / fraction from a pre-activation that already carries the factor 32 (the MLP kernels fold it into
// W1 and b1 when they load them: a power-of-two scale commutes with every rounding of the fma chain unless an
// intermediate is subnormal, |acc| < 2^-126, where the scaled chain keeps more bits of a value that is zero to
// 1e-38 either way)
SES_DEV int32_t tanh_index_scaled(float x32, float &u)
{
    const float t = min_(__builtin_fabsf(x32), SES_TANH_XMAX * SES_TANH_H_INV);
#if defined(__HIPCC__)
    u = __builtin_amdgcn_fractf(t);
#else
    u = t - __builtin_floorf(t);
#endif
    return (int32_t)t;
}

SES_DEV float tanh_eval(const TanhEntry &c, float u, float x)
{
    float p = fma_(c.c3, u, c.c2);
    p = fma_(p, u, c.c1);
    p = fma_(p, u, c.c0);
    return __builtin_copysignf(p, x);
}

SES_DEV float tanh_(const TanhEntry *tab, float x)
{
    float u;
    const int32_t i = tanh_index(x, u);
    return tanh_eval(tab[i], u, x);
}

// logistic sigmoid = 0.5 + 0.5*tanh(x/2)
SES_DEV float sigmoid_(const TanhEntry *tab, float x)
{
    return fma_(0.5f, tanh_(tab, 0.5f * x), 0.5f);
}

SES_DEV void sincos_(float x, float &s_out, float &c_out)
{
    const float k = __builtin_rintf(x * 0x1.45f306p-1f);
    float r = fma_(k, -0x1.921p+0f, x);
    r = fma_(k, -0x1.f6ap-13f, r);
    r = fma_(k, -0x1.110b46p-26f, r);
    const float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = fma_(ps, z, 8.3321608736e-3f);
    ps = fma_(ps, z, -1.6666654611e-1f);
    const float s = fma_(ps * z, r, r);
    float pc = 2.443315711809948e-5f;
    pc = fma_(pc, z, -1.388731625493765e-3f);
    pc = fma_(pc, z, 4.166664568298827e-2f);
    const float c = fma_(pc * z, z, fma_(-0.5f, z, 1.0f));
    const int32_t q = (int32_t)min_(max_(k, -1.0e9f), 1.0e9f);
    const float sv = (q & 1) ? c : s;
    const float cv = (q & 1) ? s : c;
    s_out = (q & 2) ? -sv : sv;
    c_out = ((q + 1) & 2) ? -cv : cv;
}

// sincos_ restricted to |x| <= SINCOS_SMALL_MAX < pi/4: there k = rint(x * 2/pi) is zero, the Cody-Waite
// reduction returns r = x exactly and the quadrant logic selects (s, c) unchanged, so the two polynomials on x
// itself are sincos_(x) bit for bit (x = -0 gives sin = -0 where sincos_ gives +0; equal as numbers and in
// every product or sum they enter).
constexpr float SINCOS_SMALL_MAX = 0.78f;
SES_DEV void sincos_small_(float r, float &s_out, float &c_out)
{
    const float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = fma_(ps, z, 8.3321608736e-3f);
    ps = fma_(ps, z, -1.6666654611e-1f);
    s_out = fma_(ps * z, r, r);
    float pc = 2.443315711809948e-5f;
    pc = fma_(pc, z, -1.388731625493765e-3f);
    pc = fma_(pc, z, 4.166664568298827e-2f);
    c_out = fma_(pc * z, z, fma_(-0.5f, z, 1.0f));
}

// double-precision sin/cos for the gym-order float64 CartPole (Cephes sin.c coefficients, Cody-Waite by pi/2)
SES_DEV void sincos64_(double x, double &s_out, double &c_out)
{
    const double k = __builtin_rint(x * 0x1.45f306dc9c883p-1);
    double r = __builtin_fma(k, -0x1.921fb544p+0, x);
    r = __builtin_fma(k, -0x1.0b4611a6p-34, r);
    r = __builtin_fma(k, -0x1.3198a2e037073p-69, r);
    const double z = r * r;
    double ps = 1.58962301576546568060E-10;
    ps = __builtin_fma(ps, z, -2.50507477628578072866E-8);
    ps = __builtin_fma(ps, z, 2.75573136213857245213E-6);
    ps = __builtin_fma(ps, z, -1.98412698295895385996E-4);
    ps = __builtin_fma(ps, z, 8.33333333332211858878E-3);
    ps = __builtin_fma(ps, z, -1.66666666666666307295E-1);
    const double s = __builtin_fma(ps * z, r, r);
    double pc = -1.13585365213876817300E-11;
    pc = __builtin_fma(pc, z, 2.08757008419747316778E-9);
    pc = __builtin_fma(pc, z, -2.75573141792967388112E-7);
    pc = __builtin_fma(pc, z, 2.48015872888517045348E-5);
    pc = __builtin_fma(pc, z, -1.38888888888730564116E-3);
    pc = __builtin_fma(pc, z, 4.16666666666665929218E-2);
    const double c = __builtin_fma(pc * z, z, __builtin_fma(-0.5, z, 1.0));
    const double kc = k < -1.0e9 ? -1.0e9 : (k > 1.0e9 ? 1.0e9 : k);
    const int32_t q = (int32_t)kc;
    const double sv = (q & 1) ? c : s;
    const double cv = (q & 1) ? s : c;
    s_out = (q & 2) ? -sv : sv;
    c_out = ((q + 1) & 2) ? -cv : cv;
}

// natural log, normal positive inputs
SES_DEV float log_(float x)
{
    const uint32_t u = f2u(x);
    int32_t e = (int32_t)(u >> 23) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    const bool lo = m < 0.707106781186547524f;
    e = lo ? e - 1 : e;
    m = lo ? (m + m) - 1.0f : m - 1.0f;
    const float z = m * m;
    float p = 7.0376836292e-2f;
    p = fma_(p, m, -1.1514610310e-1f);
    p = fma_(p, m, 1.1676998740e-1f);
    p = fma_(p, m, -1.2420140846e-1f);
    p = fma_(p, m, 1.4249322787e-1f);
    p = fma_(p, m, -1.6668057665e-1f);
    p = fma_(p, m, 2.0000714765e-1f);
    p = fma_(p, m, -2.4999993993e-1f);
    p = fma_(p, m, 3.3333331174e-1f);
    const float fe = (float)e;
    float y = (p * m) * z;
    y = fma_(fe, -2.12194440e-4f, y);
    y = fma_(-0.5f, z, y);
    float r = m + y;
    r = fma_(fe, 0.693359375f, r);
    return r;
}

}  // namespace ses

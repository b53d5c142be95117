/*
 * ses_oracle_math.h -- TEST INFRASTRUCTURE (oracle). Not part of the product path.
 *
 * Deterministic fp32 elementary functions used by the CPU oracle.  They are
 * built only from IEEE-754 correctly rounded primitives (+, -, *, fma, /, sqrt,
 * round-to-nearest-even, int<->float conversion, integer bit operations), so a
 * HIP kernel that evaluates the same expression tree on gfx950 produces
 * bit-identical results.  libm / hardware transcendental approximations are
 * deliberately NOT used: the reference's discrete-action rollouts
 * (/root/reference/networks/neural_network.py:29-31 argmax) are chaotic in the
 * last ulp, so the only way to have "returns identical to the CPU reference"
 * is to pin every rounding.
 *
 * Accuracy (measured by tests/test_oracle_math.py against float64 libm):
 *   o_expf <= 1 ulp on [-86, 88]; |o_tanhf - tanh| <= 1.2e-7; |o_sigmoidf - sigmoid| <= 1.2e-7;
 *   |o_sincosf - sin/cos| <= 1e-7 for |x| <= 8192; o_logf <= 1 ulp on (0, 1].
 * Polynomial coefficients are the classic single-precision Cephes minimax
 * sets (public domain, S. Moshier) -- they are data, re-used here.
 *
 * Must be compiled with -ffp-contract=off (every fma below is explicit).
 */
#ifndef SES_ORACLE_MATH_H
#define SES_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "ses_tanh_table.h"

static inline float o_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

static inline uint32_t o_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float o_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float o_minf(float a, float b) { return a < b ? a : b; }
static inline float o_maxf(float a, float b) { return a > b ? a : b; }

/* e^x for x clamped to [-86, 88]; result always a normal float. */
static inline float o_expf(float x)
{
    x = o_minf(o_maxf(x, -86.0f), 88.0f);
    const float k = __builtin_rintf(x * 0x1.715476p+0f);          /* x*log2(e), RNE */
    float r = o_fma(k, -0.693359375f, x);                          /* Cody-Waite ln2 hi */
    r = o_fma(k, 2.12194440e-4f, r);                               /* ln2 lo (hi+lo=ln2) */
    float p = 1.9875691500e-4f;
    p = o_fma(p, r, 1.3981999507e-3f);
    p = o_fma(p, r, 8.3334519073e-3f);
    p = o_fma(p, r, 4.1665795894e-2f);
    p = o_fma(p, r, 1.6666665459e-1f);
    p = o_fma(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float e = o_fma(p, r2, r);
    e = e + 1.0f;
    /* scale by 2^k with an integer add on the exponent field (k in [-124,127]) */
    const int32_t ki = (int32_t)k;
    return o_u2f(o_f2u(e) + ((uint32_t)ki << 23));
}

/* tanh: piecewise cubic on [0, 10) from the generated table (oracle/ses_tanh_table.h, DATA shared with
 * the device build), odd extension, 1.0 beyond.  u = fract(ax*32) is exact in fp32. */
static inline float o_tanhf(float x)
{
    const float t = o_minf(fabsf(x), SES_TANH_XMAX) * SES_TANH_H_INV;   /* exact scaling by 32 */
    const int32_t i = (int32_t)t;                                       /* truncation */
    const float u = t - floorf(t);                                      /* v_fract_f32: exact */
    const float *c = SES_TANH_TABLE[i];
    float p = o_fma(c[3], u, c[2]);
    p = o_fma(p, u, c[1]);
    p = o_fma(p, u, c[0]);
    return copysignf(p, x);
}

/* logistic sigmoid (torch.sigmoid inside nn.GRU) = 0.5 + 0.5*tanh(x/2) */
static inline float o_sigmoidf(float x)
{
    return o_fma(0.5f, o_tanhf(0.5f * x), 0.5f);
}

/* sin and cos together; Cody-Waite reduction by pi/2, valid to |x| ~ 8192,
 * degrades gracefully (still deterministic) beyond. */
static inline void o_sincosf(float x, float *s_out, float *c_out)
{
    const float k = __builtin_rintf(x * 0x1.45f306p-1f);          /* x * 2/pi */
    float r = o_fma(k, -0x1.921p+0f, x);
    r = o_fma(k, -0x1.f6ap-13f, r);
    r = o_fma(k, -0x1.110b46p-26f, r);
    const float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = o_fma(ps, z, 8.3321608736e-3f);
    ps = o_fma(ps, z, -1.6666654611e-1f);
    const float s = o_fma(ps * z, r, r);
    float pc = 2.443315711809948e-5f;
    pc = o_fma(pc, z, -1.388731625493765e-3f);
    pc = o_fma(pc, z, 4.166664568298827e-2f);
    float c = o_fma(pc * z, z, o_fma(-0.5f, z, 1.0f));
    /* clamp k so the int conversion is defined for huge |x| */
    const int32_t q = (int32_t)o_minf(o_maxf(k, -1.0e9f), 1.0e9f);
    const float sv = (q & 1) ? c : s;
    const float cv = (q & 1) ? s : c;
    *s_out = (q & 2) ? -sv : sv;
    *c_out = ((q + 1) & 2) ? -cv : cv;
}

/* natural log for x in (0, +inf), normal inputs only (callers pass u in [2^-33, 1]). */
static inline float o_logf(float x)
{
    uint32_t u = o_f2u(x);
    int32_t e = (int32_t)(u >> 23) - 126;                          /* frexp: m in [0.5,1) */
    float m = o_u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float p = 7.0376836292e-2f;
    p = o_fma(p, m, -1.1514610310e-1f);
    p = o_fma(p, m, 1.1676998740e-1f);
    p = o_fma(p, m, -1.2420140846e-1f);
    p = o_fma(p, m, 1.4249322787e-1f);
    p = o_fma(p, m, -1.6668057665e-1f);
    p = o_fma(p, m, 2.0000714765e-1f);
    p = o_fma(p, m, -2.4999993993e-1f);
    p = o_fma(p, m, 3.3333331174e-1f);
    const float fe = (float)e;
    float y = (p * m) * z;
    y = o_fma(fe, -2.12194440e-4f, y);
    y = o_fma(-0.5f, z, y);
    float r = m + y;
    r = o_fma(fe, 0.693359375f, r);
    return r;
}

/* double-precision sin/cos (Cephes sin.c coefficients, 3-part Cody-Waite by pi/2), |err| ~ 1.1e-16; used by the
 * gym-order float64 CartPole (physics64 mode). */
static inline void o_sincos64(double x, double *s_out, double *c_out)
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
    *s_out = (q & 2) ? -sv : sv;
    *c_out = ((q + 1) & 2) ? -cv : cv;
}

#endif /* SES_ORACLE_MATH_H */

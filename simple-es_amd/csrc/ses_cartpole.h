// ses_cartpole.h -- CartPole-v1 dynamics, one lane = one env, fp32.
//
// The reference reaches this physics through envs/gym_wrapper.py:36 (`self.env.step(action["0"])`);
// gym itself is third-party and not part of the reference tree.  These are the classic-control
// equations (euler integrator, positions advance with the old velocities), evaluated in fp32 in
// exactly the operation order of oracle/ses_oracle.c::cartpole_step.
#pragma once
#include "ses_math.h"

namespace ses {

constexpr float CP_GRAVITY = 9.8f;
constexpr float CP_FORCE_OVER_MASS = 0x1.22e8bap+3f;  // force_mag / total_mass = 10 / 1.1
constexpr float CP_PML_OVER_MASS = 0x1.745d18p-5f;    // polemass_length / total_mass = 0.05 / 1.1
constexpr float CP_DEN_C0 = 0x1.555556p-1f;           // length * 4/3
constexpr float CP_DEN_C1 = -0x1.745d18p-5f;          // -length * masspole / total_mass
constexpr float CP_TAU = 0.02f;
constexpr float CP_X_LIMIT = 2.4f;
constexpr float CP_THETA_LIMIT = 0.20943951f;  // 12 degrees
constexpr float CP_CLAMP = 1.0e4f;             // never active while an episode is alive
// The pole angle saturates at 43 degrees.  An episode ends at 12 degrees, so this never touches a live env; it
// keeps the past-terminal states of the fixed-length (termination-masked) mode inside |th| < pi/4, where the
// sin/cos argument reduction is the identity and cartpole_pre can skip it (see sincos_small_).
constexpr float CP_TH_CLAMP = 0.75f;

struct CartPoleState {
    float x, xd, th, thd;
};

SES_DEV float clamp_sym(float v, float lim) { return min_(max_(v, -lim), lim); }

// the value c in a VGPR the optimiser cannot see through (device only)
SES_DEV float register_constant(float c)
{
#if defined(__HIPCC__)
    asm volatile("" : "+v"(c));
#endif
    return c;
}

// The step is split so that a fused kernel can overlap the action-independent half (sin/cos of the pole
// angle, denominator) with the policy's LDS table reads; the arithmetic and its order are unchanged.
struct CartPolePre {
    float sn, cs, q, gsn, den;
    float rden;   // device: refined reciprocal of den for the quotient in cartpole_post (see there); host: unused
};

SES_DEV CartPolePre cartpole_pre_from(const CartPoleState &s, float sn, float cs)
{
    CartPolePre p;
    p.sn = sn;
    p.cs = cs;
    p.q = CP_PML_OVER_MASS * (s.thd * s.thd);
    p.gsn = CP_GRAVITY * p.sn;
    p.den = fma_(CP_DEN_C1, p.cs * p.cs, CP_DEN_C0);
#if defined(__HIPCC__)
    {   // first half of the IEEE division num / den of cartpole_post, the part that does not depend on the action
        const float r0 = __builtin_amdgcn_rcpf(p.den);
        p.rden = fma_(fma_(-p.den, r0, 1.0f), r0, r0);
    }
#else
    p.rden = 0.0f;
#endif
    return p;
}

// for callers that have established |th| <= SINCOS_SMALL_MAX (every state a previous step produced satisfies it)
SES_DEV CartPolePre cartpole_pre_small(const CartPoleState &s)
{
    float sn, cs;
    sincos_small_(s.th, sn, cs);
    return cartpole_pre_from(s, sn, cs);
}

SES_DEV CartPolePre cartpole_pre(const CartPoleState &s)
{
    float sn, cs;
#if defined(__HIPCC__)
    // wave-uniform choice: every lane inside |th| <= 0.78 -> the reduction-free form, bit-identical there; a
    // caller-supplied state outside it takes the general form.
    if (__builtin_expect(__ballot(!(__builtin_fabsf(s.th) <= SINCOS_SMALL_MAX)) != 0ull, 0))
        sincos_(s.th, sn, cs);
    else
        sincos_small_(s.th, sn, cs);
#else
    sincos_(s.th, sn, cs);
#endif
    return cartpole_pre_from(s, sn, cs);
}

// advances s in place; returns true when the NEW state is terminal
// th_clamp: CP_TH_CLAMP; a fused loop passes it from a register it set up once (register_constant) so that the
// clamp is a single v_med3_f32 -- with the literal the compiler emits v_max + v_min, one literal each
// num / den, correctly rounded.  den = l * (4/3 - mp cos^2 / M) lies in [0.62, 0.67] and |num| < 1e8, so the operand
// scaling and the special-case fix-up of the generic expansion (v_div_scale x2, v_div_fixup) never act: what remains is
// its Newton-Raphson core -- one step on the reciprocal (done in cartpole_pre_from, off the action's critical path), two
// on the quotient, final fma -- which yields the IEEE quotient bit for bit (the oracle divides with `/`;
// tools/fuzz_parity.py and the parity suites compare every return).
SES_DEV float cartpole_quotient(float num, const CartPolePre &p)
{
#if defined(__HIPCC__)
    const float r = p.rden;
    float q = num * r;
    q = fma_(fma_(-p.den, q, num), r, q);
    return fma_(fma_(-p.den, q, num), r, q);
#else
    return num / p.den;
#endif
}

// lim_clamp: CP_CLAMP from a register (register_constant), like th_clamp: one v_med3_f32 per clamp instead of two
// instructions with a literal each
SES_DEV float clamp_sym_reg(float v, float lim)
{
#if defined(__HIPCC__)
    return __builtin_amdgcn_fmed3f(v, -lim, lim);
#else
    return clamp_sym(v, lim);
#endif
}

SES_DEV bool cartpole_post(CartPoleState &s, const CartPolePre &p, int action, float th_clamp = CP_TH_CLAMP,
                           float lim_clamp = CP_CLAMP)
{
    const float fom = action == 1 ? CP_FORCE_OVER_MASS : -CP_FORCE_OVER_MASS;
    // temp = (F + pml*thd^2*sin)/M ; thetaacc = (g*sin - cos*temp) / (l*(4/3 - mp*cos^2/M)) ;
    // xacc = temp - pml*thetaacc*cos/M   -- constant divisions folded into multipliers, one true division
    const float temp = fma_(p.q, p.sn, fom);
    const float num = fma_(-p.cs, temp, p.gsn);
    const float thacc = cartpole_quotient(num, p);
    const float xacc = fma_(-CP_PML_OVER_MASS * thacc, p.cs, temp);
    const float nx = clamp_sym_reg(fma_(CP_TAU, s.xd, s.x), lim_clamp);
    const float nxd = clamp_sym_reg(fma_(CP_TAU, xacc, s.xd), lim_clamp);
#if defined(__HIPCC__)
    // v_med3_f32(v, -lim, lim) == min(max(v, -lim), lim) for lim >= 0, NaN included (it returns the minimum of the
    // non-NaN operands, as fmin / fmax do); the compiler makes this rewrite itself when lim is a literal
    const float nth = __builtin_amdgcn_fmed3f(fma_(CP_TAU, s.thd, s.th), -th_clamp, th_clamp);
#else
    const float nth = clamp_sym(fma_(CP_TAU, s.thd, s.th), th_clamp);
#endif
    const float nthd = clamp_sym_reg(fma_(CP_TAU, thacc, s.thd), lim_clamp);
    s.x = nx; s.xd = nxd; s.th = nth; s.thd = nthd;
    return (nx < -CP_X_LIMIT) || (nx > CP_X_LIMIT) || (nth < -CP_THETA_LIMIT) || (nth > CP_THETA_LIMIT);
}

SES_DEV bool cartpole_step(CartPoleState &s, int action)
{
    const CartPolePre p = cartpole_pre(s);
    return cartpole_post(s, p, action);
}

// The same step with the general sin/cos unconditionally.  For kernels where the physics is a small part of the
// step (the GRU rollouts): the per-step uniform branch of cartpole_pre cost them far more than the reduction it
// skips (lockstep GRU rollout 2.96 ms with this form, 4.57 ms with the branch inside the gate loop's schedule).
SES_DEV bool cartpole_step_general(CartPoleState &s, int action)
{
    float sn, cs;
    sincos_(s.th, sn, cs);
    const CartPolePre p = cartpole_pre_from(s, sn, cs);
    return cartpole_post(s, p, action);
}

// Gym-order float64 dynamics ("physics64"): the statements of gym's cartpole.py step() one by one, every
// operation a separately rounded IEEE double operation, with sincos64_ in place of libm.  Slower (f64 VALU is
// half rate) and not the benchmark path; it exists to show how much of the fp32-vs-gym deviation is precision:
// agreement of per-offspring returns with a gym-faithful float64 env rises from 94.9 % (fp32) to 99.2 %, the rest
// is libm's pow()/sin()/cos() differing from any restatement in the last ulp.
struct CartPoleState64 {
    double x, xd, th, thd;
};

SES_DEV bool cartpole_step64(CartPoleState64 &s, int action)
{
    const double gravity = 9.8, masscart = 1.0, masspole = 0.1, length = 0.5, force_mag = 10.0, tau = 0.02;
    const double total_mass = masspole + masscart, polemass_length = masspole * length;
    const double force = action == 1 ? force_mag : -force_mag;
    double sintheta, costheta;
    sincos64_(s.th, sintheta, costheta);
    const double temp = (force + ((polemass_length * (s.thd * s.thd)) * sintheta)) / total_mass;
    const double thetaacc = ((gravity * sintheta) - (costheta * temp)) /
                            (length * ((4.0 / 3.0) - ((masspole * (costheta * costheta)) / total_mass)));
    const double xacc = temp - (((polemass_length * thetaacc) * costheta) / total_mass);
    double nx = s.x + tau * s.xd, nxd = s.xd + tau * xacc, nth = s.th + tau * s.thd, nthd = s.thd + tau * thetaacc;
    const double lim = 1.0e4;
    nx = nx < -lim ? -lim : (nx > lim ? lim : nx);
    nxd = nxd < -lim ? -lim : (nxd > lim ? lim : nxd);
    nth = nth < -lim ? -lim : (nth > lim ? lim : nth);
    nthd = nthd < -lim ? -lim : (nthd > lim ? lim : nthd);
    s.x = nx; s.xd = nxd; s.th = nth; s.thd = nthd;
    const double thr = 12 * 2 * 3.141592653589793 / 360;
    return (nx < -2.4) | (nx > 2.4) | (nth < -thr) | (nth > thr);
}

}  // namespace ses

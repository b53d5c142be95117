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
constexpr float CP_MASSPOLE = 0.1f;
constexpr float CP_TOTAL_MASS = 1.1f;
constexpr float CP_LENGTH = 0.5f;
constexpr float CP_POLEMASS_LENGTH = 0.05f;
constexpr float CP_FORCE_MAG = 10.0f;
constexpr float CP_TAU = 0.02f;
constexpr float CP_X_LIMIT = 2.4f;
constexpr float CP_THETA_LIMIT = 0.20943951f;  // 12 degrees
constexpr float CP_CLAMP = 1.0e4f;             // never active while an episode is alive

struct CartPoleState {
    float x, xd, th, thd;
};

SES_DEV float clamp_sym(float v, float lim) { return min_(max_(v, -lim), lim); }

// advances s in place; returns true when the NEW state is terminal
SES_DEV bool cartpole_step(CartPoleState &s, int action)
{
    const float force = action == 1 ? CP_FORCE_MAG : -CP_FORCE_MAG;
    float sn, cs;
    sincos_(s.th, sn, cs);
    const float temp = (force + (CP_POLEMASS_LENGTH * (s.thd * s.thd)) * sn) / CP_TOTAL_MASS;
    const float thacc = (CP_GRAVITY * sn - cs * temp) /
                        (CP_LENGTH * ((4.0f / 3.0f) - (CP_MASSPOLE * (cs * cs)) / CP_TOTAL_MASS));
    const float xacc = temp - ((CP_POLEMASS_LENGTH * thacc) * cs) / CP_TOTAL_MASS;
    const float nx = clamp_sym(fma_(CP_TAU, s.xd, s.x), CP_CLAMP);
    const float nxd = clamp_sym(fma_(CP_TAU, xacc, s.xd), CP_CLAMP);
    const float nth = clamp_sym(fma_(CP_TAU, s.thd, s.th), CP_CLAMP);
    const float nthd = clamp_sym(fma_(CP_TAU, thacc, s.thd), CP_CLAMP);
    s.x = nx; s.xd = nxd; s.th = nth; s.thd = nthd;
    return (nx < -CP_X_LIMIT) | (nx > CP_X_LIMIT) | (nth < -CP_THETA_LIMIT) | (nth > CP_THETA_LIMIT);
}

}  // namespace ses

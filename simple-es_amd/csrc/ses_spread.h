// ses_spread.h -- pettingzoo MPE simple_spread dynamics, fp32, one lane group = one env.
//
// The reference reaches this env through envs/pettingzoo_wrapper.py:9,22-58 (AEC loop: every agent's
// action is set, the world advances once, observations and rewards are gathered).  pettingzoo itself is
// third-party and absent from the reference tree; the rules follow SURVEY Appendix A.3 and are evaluated
// in exactly the operation order of oracle/ses_oracle.c::spread_step / spread_obs.
// State of one env: agent positions, agent velocities, landmark positions (NA agents = NA landmarks).
#pragma once
#include "ses_math.h"

namespace ses {

constexpr float SP_DT = 0.1f;
constexpr float SP_DAMP_KEEP = 0.75f;
constexpr float SP_CONTACT_FORCE = 100.0f;
constexpr float SP_CONTACT_MARGIN = 1.0e-3f;
constexpr float SP_INV_MARGIN = 1000.0f;
constexpr float SP_DIST_MIN = 0.3f;
constexpr float SP_SENS = 5.0f;

template <int NA>
struct SpreadState {
    float ax[NA], ay[NA], vx[NA], vy[NA], lx[NA], ly[NA];
};

SES_DEV float sp_penetration(float dist)
{
    const float y = -(dist - SP_DIST_MIN) * SP_INV_MARGIN;
    const float l1p = log_(1.0f + exp_(-__builtin_fabsf(y)));
    return (max_(y, 0.0f) + l1p) * SP_CONTACT_MARGIN;
}

// observation of agent i: vel_i, pos_i, landmarks - pos_i, other agents - pos_i, comm zeros  (6*NA values)
template <int NA>
SES_DEV void spread_obs(const SpreadState<NA> &s, int i, float (&obs)[6 * NA])
{
    int o = 0;
    obs[o++] = s.vx[i]; obs[o++] = s.vy[i];
    obs[o++] = s.ax[i]; obs[o++] = s.ay[i];
#pragma unroll
    for (int k = 0; k < NA; ++k) { obs[o++] = s.lx[k] - s.ax[i]; obs[o++] = s.ly[k] - s.ay[i]; }
#pragma unroll
    for (int j = 0; j < NA; ++j)
        if (j != i) { obs[o++] = s.ax[j] - s.ax[i]; obs[o++] = s.ay[j] - s.ay[i]; }
#pragma unroll
    for (int j = 0; j < NA - 1; ++j) { obs[o++] = 0.0f; obs[o++] = 0.0f; }
}

// one world step; returns the team reward of the cycle
template <int NA>
SES_DEV float spread_step(SpreadState<NA> &s, const int (&action)[NA])
{
    float fx[NA], fy[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int a = action[i];
        fx[i] = a == 1 ? -SP_SENS : (a == 2 ? SP_SENS : 0.0f);
        fy[i] = a == 3 ? -SP_SENS : (a == 4 ? SP_SENS : 0.0f);
    }
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = a + 1; b < NA; ++b) {
            const float dx = s.ax[a] - s.ax[b], dy = s.ay[a] - s.ay[b];
            const float dist = __builtin_sqrtf(fma_(dx, dx, dy * dy));
            const float pen = sp_penetration(dist);
            const float scale = dist > 0.0f ? (SP_CONTACT_FORCE * pen) / dist : 0.0f;
            const float gx = dx * scale, gy = dy * scale;
            fx[a] = gx + fx[a]; fy[a] = gy + fy[a];
            fx[b] = fx[b] - gx; fy[b] = fy[b] - gy;
        }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const float nvx = fma_(fx[i], SP_DT, s.vx[i] * SP_DAMP_KEEP);
        const float nvy = fma_(fy[i], SP_DT, s.vy[i] * SP_DAMP_KEEP);
        s.vx[i] = nvx; s.vy[i] = nvy;
        s.ax[i] = fma_(nvx, SP_DT, s.ax[i]);
        s.ay[i] = fma_(nvy, SP_DT, s.ay[i]);
    }
    float global = 0.0f;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        float best = 0.0f;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float dx = s.ax[i] - s.lx[k], dy = s.ay[i] - s.ly[k];
            const float d = __builtin_sqrtf(fma_(dx, dx, dy * dy));
            best = i == 0 ? d : (best < d ? best : d);
        }
        global = global - best;
    }
    float team = 0.0f;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float local = 0.0f;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (j != i) {
                const float dx = s.ax[i] - s.ax[j], dy = s.ay[i] - s.ay[j];
                const float d = __builtin_sqrtf(fma_(dx, dx, dy * dy));
                local = d < SP_DIST_MIN ? local - 1.0f : local;
            }
        team = team + fma_(0.5f, global, 0.5f * local);
    }
    return team;
}

}  // namespace ses

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

// sqrtf is correctly rounded, hence monotonic, and the reward needs only two things of most of its distances:
//   (i)  the SMALLEST of the agents' distances to a landmark: the root of the smallest squared distance, bit for bit
//        (the select chain best < d ? best : d over the roots picks a value equal to it);
//   (ii) whether two agents are closer than 0.3f: sqrtf(s) < 0.3f exactly when s < SP_DIST_MIN_SQ, the smallest float
//        whose root reaches 0.3f (tests/test_host_logic.py pins the constant against sqrtf on both sides of it).
// NA landmark roots instead of NA * NA, none for the collision count instead of NA * (NA - 1): 6 instead of 18 roots
// per cycle at three agents, each ~22 instructions with the scaling of the correctly rounded form.  oracle/ses_oracle.c
// takes every root.
constexpr float SP_DIST_MIN_SQ = 0x1.70a3d8p-4f;

// contact force between two agents at offset (dx, dy): what agent a gets, agent b gets the opposite
SES_DEV void sp_pair_force(float dx, float dy, float &gx, float &gy)
{
    const float dist = __builtin_sqrtf(fma_(dx, dx, dy * dy));
    const float pen = sp_penetration(dist);
    const float scale = dist > 0.0f ? (SP_CONTACT_FORCE * pen) / dist : 0.0f;
    gx = dx * scale; gy = dy * scale;
}

#if defined(__HIPCC__)
template <int CTRL>
__device__ __forceinline__ float sp_quad_bcast(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
#endif

// one world step; returns the team reward of the cycle.
// QUADS = false: one lane = one env (the step-wise entry).  QUADS = true: every lane of a QUAD (4 adjacent lanes) holds
// the same env (the rollout kernel: 8 lanes per env = 2 quads) and all 64 lanes are active: the quad shares the work that
// is the same on all of its lanes -- lane q of the quad evaluates agent pair q's contact force (root, exp, log, divide:
// ~90 instructions) and landmark q's root, one DPP quad broadcast hands each result to the other lanes.  Same operations on
// the same operands, accumulated in the same order.
template <int NA, bool QUADS = false>
SES_DEV float spread_step(SpreadState<NA> &s, const int (&action)[NA], int qpos = 0)
{
    constexpr int NP = NA * (NA - 1) / 2;
    constexpr bool SHARE = QUADS && NP >= 2 && NP <= 4 && NA <= 4;
    float fx[NA], fy[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int a = action[i];
        fx[i] = a == 1 ? -SP_SENS : (a == 2 ? SP_SENS : 0.0f);
        fy[i] = a == 3 ? -SP_SENS : (a == 4 ? SP_SENS : 0.0f);
    }
    float gxp[NP > 0 ? NP : 1], gyp[NP > 0 ? NP : 1];
    if constexpr (SHARE) {
#if defined(__HIPCC__)
        // lane q's pair: the q-th of (0,1), (0,2), ... in the order of the loops below (a quad lane beyond the last pair
        // repeats pair 0)
        float xa = s.ax[0], ya = s.ay[0], xb = s.ax[1], yb = s.ay[1];
        {
            int p = 0;
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = a + 1; b < NA; ++b) {
                    if (p > 0) {
                        const bool mine = qpos == p;
                        xa = mine ? s.ax[a] : xa; ya = mine ? s.ay[a] : ya;
                        xb = mine ? s.ax[b] : xb; yb = mine ? s.ay[b] : yb;
                    }
                    ++p;
                }
        }
        float gx, gy;
        sp_pair_force(xa - xb, ya - yb, gx, gy);
        gxp[0] = sp_quad_bcast<0x00>(gx); gyp[0] = sp_quad_bcast<0x00>(gy);
        if constexpr (NP > 1) { gxp[1] = sp_quad_bcast<0x55>(gx); gyp[1] = sp_quad_bcast<0x55>(gy); }
        if constexpr (NP > 2) { gxp[2] = sp_quad_bcast<0xAA>(gx); gyp[2] = sp_quad_bcast<0xAA>(gy); }
        if constexpr (NP > 3) { gxp[3] = sp_quad_bcast<0xFF>(gx); gyp[3] = sp_quad_bcast<0xFF>(gy); }
#endif
    } else {
        int p = 0;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = a + 1; b < NA; ++b) {
                sp_pair_force(s.ax[a] - s.ax[b], s.ay[a] - s.ay[b], gxp[p], gyp[p]);
                ++p;
            }
    }
    {
        int p = 0;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = a + 1; b < NA; ++b) {
                fx[a] = gxp[p] + fx[a]; fy[a] = gyp[p] + fy[a];
                fx[b] = fx[b] - gxp[p]; fy[b] = fy[b] - gyp[p];
                ++p;
            }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const float nvx = fma_(fx[i], SP_DT, s.vx[i] * SP_DAMP_KEEP);
        const float nvy = fma_(fy[i], SP_DT, s.vy[i] * SP_DAMP_KEEP);
        s.vx[i] = nvx; s.vy[i] = nvy;
        s.ax[i] = fma_(nvx, SP_DT, s.ax[i]);
        s.ay[i] = fma_(nvy, SP_DT, s.ay[i]);
    }
    // squared distance of the closest agent, per landmark
    float near_sq[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        float best = 0.0f;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float dx = s.ax[i] - s.lx[k], dy = s.ay[i] - s.ly[k];
            const float sq = fma_(dx, dx, dy * dy);
            best = i == 0 ? sq : (best < sq ? best : sq);
        }
        near_sq[k] = best;
    }
    float near[NA];
    if constexpr (SHARE) {
#if defined(__HIPCC__)
        float mine = near_sq[0];
#pragma unroll
        for (int k = 1; k < NA; ++k) mine = qpos == k ? near_sq[k] : mine;
        const float root = __builtin_sqrtf(mine);
        near[0] = sp_quad_bcast<0x00>(root);
        if constexpr (NA > 1) near[1] = sp_quad_bcast<0x55>(root);
        if constexpr (NA > 2) near[2] = sp_quad_bcast<0xAA>(root);
        if constexpr (NA > 3) near[3] = sp_quad_bcast<0xFF>(root);
#endif
    } else {
#pragma unroll
        for (int k = 0; k < NA; ++k) near[k] = __builtin_sqrtf(near_sq[k]);
    }
    float global = 0.0f;
#pragma unroll
    for (int k = 0; k < NA; ++k) global = global - near[k];
    // the squared distance of a pair of agents is the same bits from either side: (-dx) * (-dx) = dx * dx
    bool close[NA][NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = i + 1; j < NA; ++j) {
            const float dx = s.ax[i] - s.ax[j], dy = s.ay[i] - s.ay[j];
            close[i][j] = close[j][i] = fma_(dx, dx, dy * dy) < SP_DIST_MIN_SQ;
        }
    float team = 0.0f;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float local = 0.0f;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (j != i) local = close[i][j] ? local - 1.0f : local;
        team = team + fma_(0.5f, global, 0.5f * local);
    }
    return team;
}

}  // namespace ses

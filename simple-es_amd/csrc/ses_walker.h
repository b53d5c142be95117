// ses_walker.h -- device build of BipedalWalker-v3: gym's bipedal_walker.py (ses_walker_env.h) on the Box2D-style
// world of ses_b2.h -- hull + four leg bodies, four revolute joints with limits and action-driven motors, leg / terrain
// contacts, 10 lidar rays, world.Step(1/50, 180, 60) per env step.
//
// The reference reaches this env through envs/gym_wrapper.py:9,36 (conf/bipedalwalker.yaml); gym + Box2D are third-party
// and absent: PARITY UNPINNED at this boundary (see ses_b2.h / ses_walker_env.h).
// Include after ses_lander.h (which defines the B2_* macros of the device build).  The terrain of an episode (200
// heights) lives in an LDS row owned by the env.
#pragma once
#include "ses_lander.h"

namespace ses {
constexpr uint64_t TAG_ENV_TERRAIN = 3ull;
}
// uniform in (-1, 1) and a raw word for terrain point i (oracle: ses_b2_oracle.cpp b2o_terrain_rand)
#define B2_TERRAIN_RAND(k0, k1, i, u, r)                                                                           \
    do {                                                                                                           \
        const uint4 w_ = ses::philox_words(((uint64_t)(k1) << 32) | (uint64_t)(k0), ses::TAG_ENV_TERRAIN, 0ull, 0u, \
                                           (uint32_t)(i));                                                         \
        u = ses::fma_(ses::u32_to_unit(w_.x), 2.0f, -1.0f);                                                        \
        r = w_.y;                                                                                                  \
    } while (0)

#include "ses_walker_env.h"

namespace ses {

constexpr int BW_TERRAIN_ROW = b2l::BW_TERRAIN_LENGTH;

struct WalkerState {
    b2l::WalkerEnv env;
    const float *ty;                         // LDS: this env's terrain heights
    b2l::WalkerPending pend;                 // between the two halves of a step
};

__device__ __forceinline__ void bw_obs(const WalkerState &s, float (&obs)[24]) { b2l::walker_obs(s.env, obs); }

// one env step as two real functions, like ll_step / ll_step_end (ses_lander.h): state in the caller's private memory,
// no wave votes inside
__device__ __attribute__((noinline)) float bw_step_end(WalkerState &s, bool &done)
{
    const b2l::WalkerTerrain terr{s.ty};
    bool d;
    const float r = b2l::walker_step_end(s.env, terr, s.pend, d);
    done = d;
    return r;
}

__device__ __attribute__((noinline)) float bw_step(WalkerState &s, float a0, float a1, float a2, float a3, bool &done)
{
    {
        B2_PHASE(12);
        b2l::WalkerEnv e = s.env;
        asm volatile("" ::: "memory");
        B2_PHASE(13);
        const b2l::WalkerTerrain terr{s.ty};
        const float act[4] = {a0, a1, a2, a3};
        b2l::WalkerPending pd;
        b2l::walker_step_begin(e, terr, act, pd);
        s.env = e;
        s.pend = pd;
    }
    return bw_step_end(s, done);
}

// reset from one row of 4 floats ([0] force uniform, [1], [2] terrain key bit patterns); ends with one no-op step.
// row: LDS row of BW_TERRAIN_ROW floats owned by this env (every lane of the env writes identical values); all lanes
// of the wave call this together.
__device__ __forceinline__ void bw_reset(WalkerState &s, const float *__restrict__ u, float *row)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    b2l::walker_terrain_heights(f2u(u[1]), f2u(u[2]), row);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    s.ty = row;
    b2l::walker_reset_state(s.env, u);
    bool done;
    (void)bw_step(s, 0.0f, 0.0f, 0.0f, 0.0f, done);
}

}  // namespace ses

// ses_lander.h -- device build of LunarLanderContinuous-v2: gym's lunar_lander.py (ses_lander_env.h) on the
// Box2D-style world of ses_b2.h -- three bodies, two revolute joints with limit + motor, polygon / terrain-edge
// contacts, world.Step(1/50, 180 velocity iterations, 60 position iterations) per env step.
//
// The reference reaches this env through envs/gym_wrapper.py:9,36 (conf/lunarlander_openai.yaml, conf/lunarlander.yaml);
// gym + Box2D are third-party and absent, so PARITY IS UNPINNED at this boundary: the two headers restate the published
// algorithms (continuous collision included) and list what deviates (own island order, float32 env arithmetic).
//
// This file supplies the B2_* macros of the device build and the small surface the rollout kernels use
// (LanderState, ll_reset, ll_obs, ll_step).  The terrain of an episode (11 smoothed heights) lives in an LDS row owned
// by the env; every lane that simulates the env passes the same row.
#pragma once
#include "ses_math.h"
#include "ses_rng.h"

#define B2_FN static __device__ __forceinline__
#define B2_NOINLINE static __device__ __attribute__((noinline))
#define B2_FN_MEMBER __device__ __forceinline__
#define B2_CONST static __device__ const
#define B2_UNROLL _Pragma("unroll")
#define B2_SINCOS(a, s, c) ses::sincos_((a), (s), (c))
#define B2_SQRT(x) __builtin_sqrtf(x)
#define B2_FLOOR(x) __builtin_floorf(x)
#define B2_RARE_PATH asm volatile("")
#define B2_CLAMP_SYM(a, lim) __builtin_amdgcn_fmed3f((a), -(lim), (lim))
#define B2_OPAQUE_PTR(p) asm volatile("" : "+v"(p)::"memory")
#ifdef SES_PHASE_TIMERS
// development build: cycles between consecutive marks of a wave by the phase that ended at the mark, accumulated in LDS
// (single-wave workgroups) and added to the totals when the wave ends (phase_flush) -- no memory operation at a mark
namespace ses {
// slots 0-15: phases (tools/walker_phases.py names them); 16: the contact rows of the velocity iterations when they are timed
// apart from the joints (-DSES_PHASE_SPLIT_VEL: two more marks per iteration); 17-19: row census of the velocity iterations
// (row slots the wave executed = union over its lanes per body / what a flat per-env list would execute = the largest lane
// total / iterations counted), see phase_rows
constexpr int PHASE_SLOTS = 24;
static __device__ unsigned long long phase_total[PHASE_SLOTS];
__device__ __forceinline__ unsigned long long *phase_lds()
{
    __shared__ unsigned long long a[4][PHASE_SLOTS + 1];       // per wave of the workgroup (<= 4)
    return a[(threadIdx.x >> 6) & 3];
}
__device__ __forceinline__ void phase_mark(int k)
{
    unsigned long long *a = phase_lds();
    const unsigned long long now = __builtin_readcyclecounter();
    const unsigned long long m = __ballot(1);
    if ((int)__lane_id() == __ffsll((long long)m) - 1) {
        if (k >= 0) a[k] += now - a[PHASE_SLOTS];
        else for (int i = 0; i < PHASE_SLOTS; ++i) a[i] = 0;
        a[PHASE_SLOTS] = now;
    }
}
// census of one world step's velocity iterations: `mask` has bit (body * slots + row) set for every contact row this lane's
// env executes, the active lanes are the envs that step
__device__ __forceinline__ void phase_rows(unsigned int mask, int iterations)
{
    unsigned long long *a = phase_lds();
    const unsigned long long m = __ballot(1);
    unsigned int uni = 0u;
    for (int bit = 0; bit < 32; ++bit) uni |= __ballot((mask >> bit) & 1u) ? (1u << bit) : 0u;
    const int mine = __popc(mask);
    int most = 0;                                              // (lanes that are not stepping take no part: reduce over the ballot)
    for (int l = 0; l < 64; ++l) {
        const int v = __builtin_amdgcn_readlane(mine, l);
        if ((m >> l) & 1ull) most = v > most ? v : most;
    }
    if ((int)__lane_id() == __ffsll((long long)m) - 1 && uni) {
        a[17] += (unsigned long long)__popc(uni) * iterations;
        a[18] += (unsigned long long)most * iterations;
        a[19] += (unsigned long long)iterations;
    }
}
__device__ __forceinline__ void phase_flush()
{
    unsigned long long *a = phase_lds();
    if (__lane_id() < PHASE_SLOTS) atomicAdd(&phase_total[__lane_id()], a[__lane_id()]);
}
}  // namespace ses
#define B2_PHASE(k) ses::phase_mark(k)
#define B2_PHASE_ROWS(mask, iters) ses::phase_rows((mask), (iters))
#endif
#define B2_F2U(f) ses::f2u(f)
// two uniforms in (-1, 1) from the episode key and the step counter (oracle: ses_b2_oracle.cpp b2o_dispersion)
#define B2_DISPERSION(k0, k1, step, d0, d1)                                                                     \
    do {                                                                                                        \
        const uint4 r_ = ses::philox_words(((uint64_t)(k1) << 32) | (uint64_t)(k0), ses::TAG_ENV_STEP, 0ull, 0u, \
                                           (uint32_t)(step));                                                   \
        d0 = ses::fma_(ses::u32_to_unit(r_.x), 2.0f, -1.0f);                                                    \
        d1 = ses::fma_(ses::u32_to_unit(r_.y), 2.0f, -1.0f);                                                    \
    } while (0)

namespace ses {
constexpr uint64_t TAG_ENV_STEP = 2ull;
}

#include "ses_lander_env.h"

namespace ses {

constexpr int LL_TERRAIN_ROW = 12;           // floats of LDS per env: 11 heights + 1 pad

struct LanderState {
    b2l::LanderEnv env;
    const float *ty;                         // LDS: this env's terrain heights
    b2l::LanderPending pend;                 // between the two halves of a step
};

__device__ __forceinline__ void ll_obs(const LanderState &s, float (&obs)[8]) { b2l::lander_obs(s.env, obs); }

// One env step; returns the reward, sets done.
// REAL functions (not inlined): a world step is ~20 000 instructions, as long as a small kernel.  Compiled once and
// called, it gets its own register allocation -- the rollout kernels keep their policy weights in registers without
// competing with the solver for them -- and every kernel shares one copy of the code.  The state lives in the caller's
// private memory.  Two functions: ll_step copies the state into registers, runs the engines and the discrete half of
// world.Step (180 solver iterations, no call inside) and writes it back; ll_step_end runs the continuous half
// (b2World::SolveTOI, the only code that calls time_of_impact) and the reward on the state where it lies -- in a step
// without an impact it touches the sweeps and a few positions.  (Both halves in one function: whatever had to survive
// the call sites, i.e. the whole world, was parked in scratch memory every step: +30 % on the C3 rollout.)
// The env code contains no wave-level operation, so the calls may sit under any divergence (finished envs simply do
// not call).
__device__ __attribute__((noinline)) float ll_step_end(LanderState &s, bool &done)
{
    const b2l::LanderTerrain terr{s.ty};
    bool d;
    const float r = b2l::lander_step_end(s.env, terr, s.pend, d);
    done = d;
    return r;
}

__device__ __attribute__((noinline)) float ll_step(LanderState &s, float a0, float a1, bool &done)
{
    {
        b2l::LanderEnv e = s.env;
        const b2l::LanderTerrain terr{s.ty};
        b2l::LanderPending pd;
        b2l::lander_step_begin(e, terr, a0, a1, pd);
        s.env = e;
        s.pend = pd;
    }
    return ll_step_end(s, done);
}

// reset from one row of 16 uniforms; like gym's reset() it ends with one no-op step.
// row: LDS row of LL_TERRAIN_ROW floats owned by this env; every lane that simulates the env passes the same row
// (they write identical values).  All lanes of the wave must call this together (wave-level LDS sync inside).
__device__ __forceinline__ void ll_reset(LanderState &s, const float *__restrict__ u, float *row)
{
    float ty[11];
    b2l::lander_terrain_heights(u, ty);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // earlier readers of the row are done
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 11; ++k) row[k] = ty[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    s.ty = row;
    b2l::lander_reset_state(s.env, u);
    bool done;
    (void)ll_step(s, 0.0f, 0.0f, done);
}

}  // namespace ses

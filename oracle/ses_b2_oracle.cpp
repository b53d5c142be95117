// ses_b2_oracle.cpp -- TEST INFRASTRUCTURE (oracle).  Host build of the Box2D-style world (ses_b2.h) and of the gym
// Box2D envs on top of it (ses_lander_env.h), with C entry points for ses_oracle.c's population rollouts and for
// the Python env objects (oracle/lander_env.py).  See those headers for what is restated and what is not pinned.
#include <math.h>
#include <stdint.h>
#include <string.h>

extern "C" {
#include "ses_oracle_math.h"
void o_philox_raw(const uint32_t *ctr, const uint32_t *key, uint32_t *out);
}

#define B2_FN static inline
#define B2_NOINLINE static __attribute__((noinline))
#define B2_FN_MEMBER inline
#define B2_CONST static const
#define B2_UNROLL
#define B2_SINCOS(a, s, c) o_sincosf((a), &(s), &(c))
#define B2_SQRT(x) sqrtf(x)
#define B2_FLOOR(x) floorf(x)
#define B2_RARE_PATH asm volatile("")
#define B2_F2U(f) o_f2u(f)

// two uniforms in (-1, 1) from the episode key and the step counter: the same Philox call as the device
// (ses_rng.h philox_words(seed, TAG_ENV_STEP, 0, 0, step)); u32_to_unit as in ses_oracle.c
static inline float b2o_u32_to_unit(uint32_t r) { return o_fma((float)r, 0x1.0p-32f, 0x1.0p-33f); }
static inline void b2o_dispersion(uint32_t key0, uint32_t key1, int step, float *d0, float *d1)
{
    const uint32_t ctr[4] = {(uint32_t)step, 0u, 0u, (uint32_t)(2ull << 24)}, key[2] = {key0, key1};
    uint32_t r[4];
    o_philox_raw(ctr, key, r);
    *d0 = o_fma(b2o_u32_to_unit(r[0]), 2.0f, -1.0f);
    *d1 = o_fma(b2o_u32_to_unit(r[1]), 2.0f, -1.0f);
}
#define B2_DISPERSION(k0, k1, step, d0, d1) b2o_dispersion((k0), (k1), (step), &(d0), &(d1))
// BipedalWalker terrain randomness for point i: the same Philox call as the device
// (ses_rng.h philox_words(seed, TAG_ENV_TERRAIN, 0, 0, i)): word 0 -> uniform in (-1, 1), word 1 raw
static inline void b2o_terrain_rand(uint32_t key0, uint32_t key1, int i, float *u, uint32_t *r)
{
    const uint32_t ctr[4] = {(uint32_t)i, 0u, 0u, (uint32_t)(3ull << 24)}, key[2] = {key0, key1};
    uint32_t w[4];
    o_philox_raw(ctr, key, w);
    *u = o_fma(b2o_u32_to_unit(w[0]), 2.0f, -1.0f);
    *r = w[1];
}
#define B2_TERRAIN_RAND(k0, k1, i, u, r) b2o_terrain_rand((k0), (k1), (i), &(u), &(r))

#include "ses_lander_env.h"
#include "ses_walker_env.h"

using namespace b2l;

struct LanderSim {
    LanderEnv env;
    float ty[11];
};

extern "C" {

int o_lander_state_size(void) { return (int)sizeof(LanderSim); }

void o_lander_obs(const void *state, float *obs)
{
    float o[8];
    lander_obs(((const LanderSim *)state)->env, o);
    memcpy(obs, o, sizeof o);
}

void o_lander_reset(void *state, const float *u16, float *obs)
{
    LanderSim *s = (LanderSim *)state;
    memset(s, 0, sizeof *s);
    lander_terrain_heights(u16, s->ty);
    LanderTerrain terr{s->ty};
    lander_reset(s->env, terr, u16);
    if (obs) o_lander_obs(state, obs);
}

float o_lander_step(void *state, float a0, float a1, float *obs, int32_t *done)
{
    LanderSim *s = (LanderSim *)state;
    LanderTerrain terr{s->ty};
    bool d;
    const float r = lander_step(s->env, terr, a0, a1, d);
    if (obs) o_lander_obs(state, obs);
    *done = d ? 1 : 0;
    return r;
}

// diagnostics for the tests: bodies [3][6] (c, a, v, w), velocity iterations of the last step, limit states, flags
void o_lander_debug(const void *state, float *bodies, int32_t *ints)
{
    const LanderSim *s = (const LanderSim *)state;
    for (int b = 0; b < 3; ++b) {
        const Body &B = s->env.w.body[b];
        const float v[6] = {B.cx, B.cy, B.a, B.vx, B.vy, B.w};
        memcpy(bodies + 6 * b, v, sizeof v);
    }
    ints[0] = 0;
    ints[1] = s->env.w.joint[0].state;
    ints[2] = s->env.w.joint[1].state;
    ints[3] = s->env.w.game_over;
    ints[4] = s->env.w.awake;
    ints[5] = s->env.w.ground_contact[1];
    ints[6] = s->env.w.ground_contact[2];
    int touching = 0;
    for (int b = 0; b < 3; ++b)
        for (int k = 0; k < 2; ++k) touching += b >= 1 ? s->env.w.mf[b - 1][k].count : 0;
    ints[7] = touching;
}

// time of impact of a lander body's polygon (0 = hull, 1 / 2 = legs) swept from (c0, a0) to (c1, a1) against the edge
// e = (x1, y1, x2, y2); returns the b2TOIOutput state (0 failed, 1 overlapped, 2 touching, 3 separated), *t = fraction
int o_toi_probe(int body, const float *e, const float *c0a0, const float *c1a1, float *t)
{
    ToiPair pr;
    pr.ex[0] = e[0]; pr.ey[0] = e[1]; pr.ex[1] = e[2]; pr.ey[1] = e[3];
    pr.P = &LANDER_POLY[body];
    Sweep sw;
    sw.c0x = c0a0[0]; sw.c0y = c0a0[1]; sw.a0 = c0a0[2];
    sw.cx = c1a1[0]; sw.cy = c1a1[1]; sw.a = c1a1[2];
    sw.alpha0 = 0.0f;
    return time_of_impact(pr, sw, LANDER_BODY[body], *t);
}

// ---- BipedalWalker-v3 ----
struct WalkerSim {
    WalkerEnv env;
    float ty[BW_TERRAIN_LENGTH];
};

int o_walker_state_size(void) { return (int)sizeof(WalkerSim); }

void o_walker_obs(const void *state, float *obs)
{
    float o[24];
    walker_obs(((const WalkerSim *)state)->env, o);
    memcpy(obs, o, sizeof o);
}

// init row: [0] initial-force uniform, [1], [2] the bit patterns of the terrain key, [3] unused
void o_walker_reset(void *state, const float *u4, float *obs)
{
    WalkerSim *s = (WalkerSim *)state;
    memset(s, 0, sizeof *s);
    walker_terrain_heights(o_f2u(u4[1]), o_f2u(u4[2]), s->ty);
    WalkerTerrain terr{s->ty};
    walker_reset(s->env, terr, u4);
    if (obs) o_walker_obs(state, obs);
}

float o_walker_step(void *state, const float *action4, float *obs, int32_t *done)
{
    WalkerSim *s = (WalkerSim *)state;
    WalkerTerrain terr{s->ty};
    const float a[4] = {action4[0], action4[1], action4[2], action4[3]};
    bool d;
    const float r = walker_step(s->env, terr, a, d);
    if (obs) o_walker_obs(state, obs);
    *done = d ? 1 : 0;
    return r;
}

// the float32 world's tables for the five walker bodies: props[5][5] = mass, inertia about the centre of mass, local centre x, y,
// friction mixed with the terrain's (tests/test_oracle_walker.py compares them with what oracle/walker64.c derives from gym's polygons)
void o_walker_body_props(float *props)
{
    for (int b = 0; b < 5; ++b) {
        const BodyDef &d = WALKER_BODY[b];
        const float v[5] = {1.0f / d.inv_mass, 1.0f / d.inv_i, d.lcx, d.lcy, d.friction};
        memcpy(props + 5 * b, v, sizeof v);
    }
}

// diagnostics for the tests: bodies [5][6] (c, a, v, w), terrain [200], flags {game_over, contact points, limit states x4,
// touching manifolds of each leg body x4}
void o_walker_debug(const void *state, float *bodies, float *terrain, int32_t *ints)
{
    const WalkerSim *s = (const WalkerSim *)state;
    for (int b = 0; b < 5; ++b) {
        const Body &B = s->env.w.body[b];
        const float v[6] = {B.cx, B.cy, B.a, B.vx, B.vy, B.w};
        memcpy(bodies + 6 * b, v, sizeof v);
    }
    memcpy(terrain, s->ty, sizeof s->ty);
    ints[0] = s->env.w.game_over;
    int touching = 0;
    for (int b = 0; b < 4; ++b)
        for (int k = 0; k < 4; ++k) touching += s->env.w.mf[b][k].count;
    ints[1] = touching;
    for (int j = 0; j < 4; ++j) ints[2 + j] = s->env.w.joint[j].state;
    for (int b = 0; b < 4; ++b) {
        ints[6 + b] = 0;
        for (int k = 0; k < 4; ++k) ints[6 + b] += s->env.w.mf[b][k].count > 0;
    }
}

}  // extern "C"

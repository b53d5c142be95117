// ses_lander.h -- LunarLanderContinuous "lite": the reduced rigid-body lander defined in
// oracle/ses_oracle.c (ll_step / ll_reset / ll_obs), transliterated operation by operation.
//
// The reference reaches LunarLanderContinuous-v2 through envs/gym_wrapper.py:9,36 (conf/lunarlander_openai.yaml);
// gym + Box2D are third-party and absent, so this env is the build's own definition: gym's constants, engine
// model with per-step dispersion noise, terrain generator, observation, reward shaping and termination rules on
// ONE rigid body with two leg-tip contacts (sequential impulses).  PARITY UNPINNED at this boundary.
#pragma once
#include "ses_math.h"
#include "ses_rng.h"

namespace ses {

constexpr float LL_SCALE = 30.0f;
constexpr float LL_DT = 0.02f;
constexpr float LL_W = 20.0f;
constexpr float LL_H = 400.0f / 30.0f;
constexpr float LL_HELIPAD_Y = 400.0f / 30.0f / 4.0f;
constexpr float LL_MAIN_POWER = 13.0f;
constexpr float LL_SIDE_POWER = 0.6f;
constexpr float LL_INV_MASS = 0x1.9cfee8p-3f;
constexpr float LL_INV_INERTIA = 0x1.18f758p+0f;
constexpr float LL_TIP_X = 0x1.e72fccp-1f;
constexpr float LL_TIP_Y = -0x1.13c8b4p-1f;
constexpr float LL_FRICTION = 0.1414f;
constexpr float LL_BAUMGARTE = 0.2f;
constexpr float LL_SLOP = 0.005f;
constexpr float LL_CRASH_SPEED = 3.0f;
constexpr float LL_SLEEP_V = 0.05f;
constexpr int LL_SLEEP_STEPS = 25;
constexpr uint64_t TAG_ENV_STEP = 2ull;

// The terrain of an episode never changes, so reset() tabulates its 10 segments once -- {left height, rise,
// normal x, normal y} per segment, with the very operations ll_terrain used to repeat at each of its 8 calls per step
// (a sqrt and a division among them) -- into an LDS row owned by the env; ll_terrain is then one ds_read_b128.
constexpr int LL_SEGMENTS = 10;

struct LanderState {
    float x, y, vx, vy, ang, om;
    float prev_shaping;
    int has_prev, sleep, leg0, leg1;
    const float4 *seg;       // LDS: this env's segment table [LL_SEGMENTS]
    float o[6];              // observation components 0..5 of the current state (the step computes them for the
                             // shaping reward; the policy reads the same values: five divisions saved per step)
    uint32_t key0, key1;
    int step;
};

__device__ __forceinline__ float4 ll_segment(float y0, float y1)
{
    const float d = y1 - y0;
    const float slope = d * 0.5f;
    const float inv = 1.0f / __builtin_sqrtf(fma_(slope, slope, 1.0f));
    return float4{y0, d, -slope * inv, inv};
}

__device__ __forceinline__ void ll_terrain(float x, const float4 *seg, float &h, float &nx, float &ny)
{
    float fk = __builtin_floorf(x * 0.5f);
    fk = min_(max_(fk, 0.0f), 9.0f);
    const int k = (int)fk;
    const float t = (x - 2.0f * fk) * 0.5f;
    const float4 e = seg[k];
    h = fma_(e.y, t, e.x);
    nx = e.z;
    ny = e.w;
}

__device__ __forceinline__ void ll_obs(const LanderState &s, float (&obs)[8])
{
#pragma unroll
    for (int k = 0; k < 6; ++k) obs[k] = s.o[k];
    obs[6] = s.leg0 ? 1.0f : 0.0f;
    obs[7] = s.leg1 ? 1.0f : 0.0f;
}

// the observation from the state variables (oracle/ses_oracle.c ll_obs)
__device__ __forceinline__ void ll_obs_compute(const LanderState &s, float (&obs)[8])
{
    obs[0] = (s.x - LL_W * 0.5f) / (LL_W * 0.5f);
    obs[1] = (s.y - (LL_HELIPAD_Y + 18.0f / LL_SCALE)) / (LL_H * 0.5f);
    obs[2] = s.vx * (LL_W * 0.5f) / 50.0f;
    obs[3] = s.vy * (LL_H * 0.5f) / 50.0f;
    obs[4] = s.ang;
    obs[5] = 20.0f * s.om / 50.0f;
    obs[6] = s.leg0 ? 1.0f : 0.0f;
    obs[7] = s.leg1 ? 1.0f : 0.0f;
}

// one env step; returns the reward, sets done
__device__ __forceinline__ float ll_step(LanderState &s, float a0, float a1, bool &done)
{
    const uint64_t seed = ((uint64_t)s.key1 << 32) | (uint64_t)s.key0;
    const uint4 r = philox_words(seed, TAG_ENV_STEP, 0ull, 0u, (uint32_t)s.step);
    const float d0 = fma_(u32_to_unit(r.x), 2.0f, -1.0f) / LL_SCALE;
    const float d1 = fma_(u32_to_unit(r.y), 2.0f, -1.0f) / LL_SCALE;
    s.step += 1;
    float sn, cs;
    sincos_(s.ang, sn, cs);
    const float tipx = sn, tipy = cs, sidex = -cs, sidey = sn;
    float m_power = 0.0f, s_power = 0.0f;
    a0 = min_(max_(a0, -1.0f), 1.0f);
    a1 = min_(max_(a1, -1.0f), 1.0f);
    if (a0 > 0.0f) {
        m_power = (min_(max_(a0, 0.0f), 1.0f) + 1.0f) * 0.5f;
        const float ox = fma_(tipx, 4.0f / LL_SCALE + 2.0f * d0, sidex * d1);
        const float oy = -(tipy * (4.0f / LL_SCALE + 2.0f * d0)) - sidey * d1;
        const float jx = -ox * LL_MAIN_POWER * m_power, jy = -oy * LL_MAIN_POWER * m_power;
        s.vx = fma_(jx, LL_INV_MASS, s.vx);
        s.vy = fma_(jy, LL_INV_MASS, s.vy);
        s.om = fma_(ox * jy - oy * jx, LL_INV_INERTIA, s.om);
    }
    if (__builtin_fabsf(a1) > 0.5f) {
        const float dir = a1 > 0.0f ? 1.0f : -1.0f;
        s_power = min_(max_(__builtin_fabsf(a1), 0.5f), 1.0f);
        const float lat = fma_(3.0f, d1, dir * (12.0f / LL_SCALE));
        const float ox = fma_(tipx, d0, sidex * lat);
        const float oy = -(tipy * d0) - sidey * lat;
        const float rx = ox - tipx * (17.0f / LL_SCALE), ry = oy + tipy * (14.0f / LL_SCALE);
        const float jx = -ox * LL_SIDE_POWER * s_power, jy = -oy * LL_SIDE_POWER * s_power;
        s.vx = fma_(jx, LL_INV_MASS, s.vx);
        s.vy = fma_(jy, LL_INV_MASS, s.vy);
        s.om = fma_(rx * jy - ry * jx, LL_INV_INERTIA, s.om);
    }
    s.vy = fma_(-10.0f, LL_DT, s.vy);

    float crx[2], cry[2], cnx[2], cny[2], cpen[2], ln[2] = {0.0f, 0.0f}, lt[2] = {0.0f, 0.0f};
    bool active[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float bx = i == 0 ? LL_TIP_X : -LL_TIP_X, by = LL_TIP_Y;
        crx[i] = bx * cs - by * sn;
        cry[i] = bx * sn + by * cs;
        float h;
        ll_terrain(s.x + crx[i], s.seg, h, cnx[i], cny[i]);
        cpen[i] = h - (s.y + cry[i]);
        active[i] = cpen[i] >= 0.0f;
    }
    bool crash = false;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float vn = fma_(s.vx - s.om * cry[i], cnx[i], fma_(s.om, crx[i], s.vy) * cny[i]);
        crash = crash | (active[i] & (vn < -LL_CRASH_SPEED));
    }
    constexpr float HULL[6][2] = {{-14.0f / 30.0f, 17.0f / 30.0f}, {-17.0f / 30.0f, 0.0f}, {-17.0f / 30.0f, -10.0f / 30.0f},
                                  {17.0f / 30.0f, -10.0f / 30.0f}, {17.0f / 30.0f, 0.0f}, {14.0f / 30.0f, 17.0f / 30.0f}};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float rx = HULL[k][0] * cs - HULL[k][1] * sn, ry = HULL[k][0] * sn + HULL[k][1] * cs;
        float h, nx, ny;
        ll_terrain(s.x + rx, s.seg, h, nx, ny);
        crash = crash | (h - (s.y + ry) >= 0.0f);
    }
    if (active[0] | active[1]) {
        for (int it = 0; it < 8; ++it) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!active[i]) continue;
                const float rx = crx[i], ry = cry[i], nx = cnx[i], ny = cny[i];
                float vpx = s.vx - s.om * ry, vpy = fma_(s.om, rx, s.vy);
                const float rn = rx * ny - ry * nx;
                const float kn = fma_(rn * rn, LL_INV_INERTIA, LL_INV_MASS);
                const float bias = (LL_BAUMGARTE / LL_DT) * max_(cpen[i] - LL_SLOP, 0.0f);
                float lam = -(fma_(vpx, nx, vpy * ny) - bias) / kn;
                const float nl = max_(ln[i] + lam, 0.0f);
                lam = nl - ln[i];
                ln[i] = nl;
                s.vx = fma_(lam * nx, LL_INV_MASS, s.vx);
                s.vy = fma_(lam * ny, LL_INV_MASS, s.vy);
                s.om = fma_(rn * lam, LL_INV_INERTIA, s.om);
                vpx = s.vx - s.om * ry;
                vpy = fma_(s.om, rx, s.vy);
                const float tx = ny, tyv = -nx;
                const float rt = rx * tyv - ry * tx;
                const float kt = fma_(rt * rt, LL_INV_INERTIA, LL_INV_MASS);
                float lamt = -fma_(vpx, tx, vpy * tyv) / kt;
                const float lim = LL_FRICTION * ln[i];
                const float ntl = min_(max_(lt[i] + lamt, -lim), lim);
                lamt = ntl - lt[i];
                lt[i] = ntl;
                s.vx = fma_(lamt * tx, LL_INV_MASS, s.vx);
                s.vy = fma_(lamt * tyv, LL_INV_MASS, s.vy);
                s.om = fma_(rt * lamt, LL_INV_INERTIA, s.om);
            }
        }
    }
    s.x = fma_(s.vx, LL_DT, s.x);
    s.y = fma_(s.vy, LL_DT, s.y);
    s.ang = fma_(s.om, LL_DT, s.ang);
    s.leg0 = active[0];
    s.leg1 = active[1];
    const float speed2 = fma_(s.vx, s.vx, s.vy * s.vy);
    const bool still = active[0] & active[1] & (speed2 < LL_SLEEP_V * LL_SLEEP_V) & (__builtin_fabsf(s.om) < LL_SLEEP_V);
    s.sleep = still ? s.sleep + 1 : 0;

    float obs[8];
    ll_obs_compute(s, obs);
#pragma unroll
    for (int k = 0; k < 6; ++k) s.o[k] = obs[k];
    const float shaping = -100.0f * __builtin_sqrtf(fma_(obs[0], obs[0], obs[1] * obs[1])) -
                          100.0f * __builtin_sqrtf(fma_(obs[2], obs[2], obs[3] * obs[3])) -
                          100.0f * __builtin_fabsf(obs[4]) + 10.0f * obs[6] + 10.0f * obs[7];
    float reward = s.has_prev ? shaping - s.prev_shaping : 0.0f;
    s.prev_shaping = shaping;
    s.has_prev = 1;
    reward = reward - m_power * 0.30f;
    reward = reward - s_power * 0.03f;
    done = false;
    if (crash | (__builtin_fabsf(obs[0]) >= 1.0f)) { done = true; reward = -100.0f; }
    if (s.sleep >= LL_SLEEP_STEPS) { done = true; reward = 100.0f; }
    return reward;
}

// reset from one row of 16 uniforms; like gym's reset() it ends with one no-op step.
// seg_row: LDS row of LL_SEGMENTS float4 owned by this env; every lane that simulates the env passes the same row
// (they write identical values).  All lanes of the wave must call this together (wave-level LDS sync inside).
__device__ __forceinline__ void ll_reset(LanderState &s, const float *__restrict__ u, float4 *seg_row)
{
    float height[12], ty[11];
#pragma unroll
    for (int i = 0; i < 12; ++i) height[i] = u[2 + i] * (LL_H * 0.5f);
#pragma unroll
    for (int i = 3; i <= 7; ++i) height[i] = LL_HELIPAD_Y;
#pragma unroll
    for (int i = 0; i < 11; ++i) ty[i] = 0.33f * ((height[i == 0 ? 11 : i - 1] + height[i]) + height[i + 1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // earlier readers of the row are done
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < LL_SEGMENTS; ++k) seg_row[k] = ll_segment(ty[k], ty[k + 1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    s.seg = seg_row;
    s.x = LL_W * 0.5f;
    s.y = LL_H;
    s.vx = (fma_(u[0], 2.0f, -1.0f) * 1000.0f) * LL_INV_MASS * LL_DT;
    s.vy = (fma_(u[1], 2.0f, -1.0f) * 1000.0f) * LL_INV_MASS * LL_DT;
    s.ang = 0.0f;
    s.om = 0.0f;
    s.prev_shaping = 0.0f;
    s.has_prev = 0;
    s.sleep = 0;
    s.leg0 = 0;
    s.leg1 = 0;
    s.key0 = f2u(u[14]);
    s.key1 = f2u(u[15]);
    s.step = 0;
    bool done;
    (void)ll_step(s, 0.0f, 0.0f, done);
}

}  // namespace ses

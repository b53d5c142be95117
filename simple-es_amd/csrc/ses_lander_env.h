// ses_lander_env.h -- LunarLanderContinuous-v2 as gym 0.18-0.21 defines it (gym/envs/box2d/lunar_lander.py; reached by
// the reference through envs/gym_wrapper.py:9,36 with conf/lunarlander_openai.yaml) on top of the Box2D-style world of
// ses_b2.h: the lander hull (density 5) and two legs (density 1) on revolute joints with limits and a 40 N*m motor
// "spring", ten terrain edges, world.Step(1/50, 6*30, 2*30) per env step, engines as impulses at the nozzle positions
// with per-step dispersion noise, gym's observation / shaping reward / termination rules.
//
// One text compiled twice, like ses_b2.h (see there): for gfx950 by the product, for the host by oracle/ses_b2_oracle.cpp
// (test infrastructure, -I simple-es_amd/csrc).
// gym and Box2D are third-party and absent here: PARITY UNPINNED at this boundary (see ses_b2.h for the list of
// deviations).  Env-level deviations: gym evaluates the engine geometry and the observation in Python doubles, here
// everything is float32; np_random is replaced by a row of 16 uniforms per episode ([0,1] initial force, [2..13] terrain
// heights, [14,15] the bit patterns of the key of the Philox stream that supplies the per-step dispersion noise); the
// engine-exhaust particles (decorative bodies that only touch the ground) are not simulated.
#pragma once
#include "ses_b2.h"

namespace b2l {

constexpr float LL_SCALE = 30.0f;
constexpr float LL_FPS = 50.0f;
constexpr float LL_DT = 0.02f;                       // float32(1.0 / FPS)
constexpr float LL_W = 20.0f;                        // VIEWPORT_W / SCALE
constexpr float LL_H = 400.0f / 30.0f;               // VIEWPORT_H / SCALE
constexpr float LL_HELIPAD_Y = 400.0f / 30.0f / 4.0f;
constexpr float LL_MAIN_POWER = 13.0f;
constexpr float LL_SIDE_POWER = 0.6f;
constexpr float LL_LEG_AWAY = 20.0f / 30.0f;
constexpr float LL_LEG_DOWN = 18.0f / 30.0f;
constexpr float LL_LEG_SPRING_TORQUE = 40.0f;
constexpr int LL_SEGMENTS = 10;

B2_CONST JointDef LANDER_JOINT[2] = {
    // leg i = -1: bodyA = lander, localAnchorA (0,0), localAnchorB (i*LEG_AWAY, LEG_DOWN)/SCALE, limits [0.9 - 0.5, 0.9]
    {0, 1, 0.0f, 0.0f, -LL_LEG_AWAY, LL_LEG_DOWN, 0.4f, 0.9f},
    // leg i = +1: limits [-0.9, -0.9 + 0.5]
    {0, 2, 0.0f, 0.0f, LL_LEG_AWAY, LL_LEG_DOWN, -0.9f, -0.4f},
};

struct LanderDef {
    static constexpr int NB = 3, NJ = 2, NSLOT = 2, FIRST_SOLVED = 1, VEL_ITERS = 6 * 30, POS_ITERS = 2 * 30;
    static constexpr bool PACK_MANIFOLDS = false;    // 2 x 2 slots: the solver runs over them as they are
    static constexpr bool CONTINUOUS = true;         // b2World::SolveTOI against the terrain (ses_b2.h)
    // The velocity iteration is a deterministic map of (velocities, accumulated impulses): once an iteration returns them
    // bit for bit, every later one does.  Measured on the CPU build (first-generation C3 policies, 48 778 steps in flight):
    // 58 % of the steps are at such a fixed point after <= 6 of the 180 iterations (limits inactive, motors saturated),
    // 39 % never reach one; with contacts 3 % by iteration 20.  One comparison, after the iteration with this index.
    // (-DB2_RUN_ALL_ITERATIONS: a CHECKER build of the oracle that takes neither this exit nor the sub-step's, ses_b2.h --
    // tests/test_oracle_lander.py holds the two builds to the same bits.)
#ifdef B2_RUN_ALL_ITERATIONS
    static constexpr int VEL_FIXED_POINT_CHECK = -1;
#else
    static constexpr int VEL_FIXED_POINT_CHECK = 7;
#endif
    static constexpr float GRAVITY_Y = -10.0f;
    B2_FN const Poly *poly() { return LANDER_POLY; }
    B2_FN const BodyDef *body() { return LANDER_BODY; }
    B2_FN const JointDef *joint() { return LANDER_JOINT; }
};

struct LanderTerrain {
    const float *ty;                                 // B2_TERRAIN_MEM: 11 smoothed heights at x = 0, 2, ..., 20
    B2_FN_MEMBER int n_edges() const { return LL_SEGMENTS; }
    B2_FN_MEMBER int index_of(float x) const { return (int)B2_FLOOR(x * 0.5f); }
    B2_FN_MEMBER void edge(int k, float &x1, float &y1, float &x2, float &y2) const
    {
        x1 = 2.0f * (float)k; y1 = ty[k];
        x2 = 2.0f * (float)(k + 1); y2 = ty[k + 1];
    }
};

struct LanderEnv {
    World<LanderDef> w;
    float prev_shaping;
    int has_prev;
    float posx, posy;                                // lander.position (body origin) after the last world step
    uint32_t key0, key1;
    int step;
};

B2_FN void lander_terrain_heights(const float *u, float *ty /*[11]*/)
{
    float height[12];
    B2_UNROLL
    for (int i = 0; i < 12; ++i) height[i] = u[2 + i] * (LL_H * 0.5f);
    B2_UNROLL
    for (int i = 3; i <= 7; ++i) height[i] = LL_HELIPAD_Y;
    B2_UNROLL
    for (int i = 0; i < 11; ++i) ty[i] = 0.33f * ((height[i == 0 ? 11 : i - 1] + height[i]) + height[i + 1]);
}

B2_FN void lander_obs(const LanderEnv &e, float (&obs)[8])
{
    const Body &L = e.w.body[0];
    obs[0] = (e.posx - LL_W * 0.5f) / (LL_W * 0.5f);
    obs[1] = (e.posy - (LL_HELIPAD_Y + LL_LEG_DOWN)) / (LL_H * 0.5f);
    obs[2] = L.vx * (LL_W * 0.5f) / LL_FPS;
    obs[3] = L.vy * (LL_H * 0.5f) / LL_FPS;
    obs[4] = L.a;
    obs[5] = 20.0f * L.w / LL_FPS;
    obs[6] = e.w.ground_contact[1] ? 1.0f : 0.0f;
    obs[7] = e.w.ground_contact[2] ? 1.0f : 0.0f;
}

// b2Body::ApplyLinearImpulse on the lander
B2_FN void lander_impulse(LanderEnv &e, float jx, float jy, float px, float py)
{
    Body &L = e.w.body[0];
    const BodyDef &bd = LANDER_BODY[0];
    L.vx += bd.inv_mass * jx;
    L.vy += bd.inv_mass * jy;
    L.w += bd.inv_i * ((px - L.cx) * jy - (py - L.cy) * jx);
}

// one env step (lunar_lander.py step()); returns the reward, sets done.  In two halves (ses_b2.h, world_step_toi, says
// why): lander_step_begin = engines + the discrete half of world.Step, lander_step_end = its continuous half + reward.
struct LanderPending {               // what the first half hands to the second
    Sweep sw[3];
    float m_power, s_power;
};

template <class T>
B2_FN void lander_step_begin(LanderEnv &e, const T &terr, float a0, float a1, LanderPending &pd)
{
    float d0, d1;
    B2_DISPERSION(e.key0, e.key1, e.step, d0, d1);   // two uniforms in (-1, 1), divided by SCALE below
    d0 = d0 / LL_SCALE;
    d1 = d1 / LL_SCALE;
    e.step += 1;
    a0 = b2clamp(a0, -1.0f, 1.0f);
    a1 = b2clamp(a1, -1.0f, 1.0f);
    float sn, cs;
    B2_SINCOS(e.w.body[0].a, sn, cs);
    const float tipx = sn, tipy = cs, sidex = -cs, sidey = sn;
    float m_power = 0.0f, s_power = 0.0f;
    if (a0 > 0.0f) {
        m_power = (b2clamp(a0, 0.0f, 1.0f) + 1.0f) * 0.5f;
        const float ox = tipx * (4.0f / LL_SCALE + 2.0f * d0) + sidex * d1;
        const float oy = -tipy * (4.0f / LL_SCALE + 2.0f * d0) - sidey * d1;
        lander_impulse(e, -ox * LL_MAIN_POWER * m_power, -oy * LL_MAIN_POWER * m_power, e.posx + ox, e.posy + oy);
    }
    if (b2abs(a1) > 0.5f) {
        const float dir = a1 > 0.0f ? 1.0f : -1.0f;
        s_power = b2clamp(b2abs(a1), 0.5f, 1.0f);
        const float lat = 3.0f * d1 + dir * (12.0f / LL_SCALE);
        const float ox = tipx * d0 + sidex * lat;
        const float oy = -tipy * d0 - sidey * lat;
        lander_impulse(e, -ox * LL_SIDE_POWER * s_power, -oy * LL_SIDE_POWER * s_power,
                       e.posx + ox - tipx * (17.0f / LL_SCALE), e.posy + oy + tipy * (14.0f / LL_SCALE));
    }
    world_step_discrete(e.w, terr, LL_DT, pd.sw);
    pd.m_power = m_power; pd.s_power = s_power;
}

template <class T>
B2_FN float lander_step_end(LanderEnv &e, const T &terr, LanderPending &pd, bool &done)
{
    world_step_toi(e.w, terr, LL_DT, pd.sw);
    const float m_power = pd.m_power, s_power = pd.s_power;
    {
        Xf x;
        xf_of(e.w.body[0], LANDER_BODY[0], x);
        e.posx = x.px; e.posy = x.py;
    }
    float obs[8];
    lander_obs(e, obs);
    const float shaping = -100.0f * B2_SQRT(obs[0] * obs[0] + obs[1] * obs[1]) - 100.0f * B2_SQRT(obs[2] * obs[2] + obs[3] * obs[3]) -
                          100.0f * b2abs(obs[4]) + 10.0f * obs[6] + 10.0f * obs[7];
    float reward = e.has_prev ? shaping - e.prev_shaping : 0.0f;
    e.prev_shaping = shaping;
    e.has_prev = 1;
    reward = reward - m_power * 0.30f;
    reward = reward - s_power * 0.03f;
    done = false;
    if (e.w.game_over || b2abs(obs[0]) >= 1.0f) { done = true; reward = -100.0f; }
    if (!e.w.awake) { done = true; reward = 100.0f; }
    return reward;
}

template <class T>
B2_FN float lander_step(LanderEnv &e, const T &terr, float a0, float a1, bool &done)
{
    LanderPending pd;
    lander_step_begin(e, terr, a0, a1, pd);
    return lander_step_end(e, terr, pd, done);
}

// reset from one row of 16 uniforms (the caller has tabulated the terrain with lander_terrain_heights).  Like gym's
// reset() it must be followed by one no-op step: lander_reset = lander_reset_state + lander_step(0, 0).
B2_FN void lander_reset_state(LanderEnv &e, const float *u)
{
    World<LanderDef> &w = e.w;
    const float initial_y = LL_H;
    B2_UNROLL
    for (int b = 0; b < 3; ++b) {
        const float i = b == 1 ? -1.0f : 1.0f;
        const float ox = b == 0 ? LL_W * 0.5f : LL_W * 0.5f - i * LL_LEG_AWAY;
        const float ang = b == 0 ? 0.0f : i * 0.05f;
        // b2Body constructor: m_sweep.c = b2Mul(m_xf, localCenter); the legs' local centre is (0, 0)
        w.body[b].cx = ox + LANDER_BODY[b].lcx; w.body[b].cy = initial_y + LANDER_BODY[b].lcy;
        w.body[b].a = ang;
        w.body[b].vx = 0.0f; w.body[b].vy = 0.0f; w.body[b].w = 0.0f;
        w.xf[b].s = 0.0f; w.xf[b].c = 1.0f; w.xf[b].px = ox; w.xf[b].py = initial_y;   // recomputed by the first Collide
        w.sleep_time[b] = 0.0f;
        w.ground_contact[b] = false;
        B2_UNROLL
        for (int s = 0; s < LanderDef::NSLOT && b >= LanderDef::FIRST_SOLVED; ++s) {
            Manifold &m = w.mf[b - LanderDef::FIRST_SOLVED][s];
            m.edge = -1; m.count = 0; m.type = 0;
            m.lnx = 0.0f; m.lny = 0.0f; m.lpx = 0.0f; m.lpy = 0.0f;
            B2_UNROLL
            for (int i2 = 0; i2 < 2; ++i2) { m.px[i2] = 0.0f; m.py[i2] = 0.0f; m.id[i2] = 0u; m.ni[i2] = 0.0f; m.ti[i2] = 0.0f; }
        }
    }
    B2_UNROLL
    for (int j = 0; j < 2; ++j) {
        Joint &J = w.joint[j];
        J.ix = 0.0f; J.iy = 0.0f; J.iz = 0.0f; J.im = 0.0f;
        J.state = LIMIT_INACTIVE;
        J.motor_speed = j == 0 ? -0.3f : 0.3f;       // +0.3 * i
        J.max_torque = LL_LEG_SPRING_TORQUE;
    }
    w.game_over = false; w.awake = true;
    w.fx = 2000.0f * u[0] - 1000.0f;                 // np_random.uniform(-INITIAL_RANDOM, INITIAL_RANDOM)
    w.fy = 2000.0f * u[1] - 1000.0f;
    e.prev_shaping = 0.0f;
    e.has_prev = 0;
    e.posx = LL_W * 0.5f; e.posy = initial_y;
    e.key0 = B2_F2U(u[14]);
    e.key1 = B2_F2U(u[15]);
    e.step = 0;
}

template <class T>
B2_FN void lander_reset(LanderEnv &e, const T &terr, const float *u)
{
    lander_reset_state(e, u);
    bool done;
    (void)lander_step(e, terr, 0.0f, 0.0f, done);
}

}  // namespace b2l

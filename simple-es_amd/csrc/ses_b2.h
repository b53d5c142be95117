// ses_b2.h -- a small rigid-body world in the manner of Box2D 2.3.0 (the engine behind gym's Box2D envs, reached
// by the reference through envs/gym_wrapper.py:9,36 for LunarLanderContinuous-v2 / BipedalWalker-v3).
//
// ONE TEXT, COMPILED TWICE: for gfx950 by csrc/ses_lander.h / ses_walker.h (the product) and for the host by
// oracle/ses_b2_oracle.cpp (test infrastructure; oracle/Makefile adds -I simple-es_amd/csrc).  The including file supplies
// the B2_* macros (function qualifiers, sincos, sqrt).  One text, so that every float operation happens in the same
// order on both sides and the GPU rollouts can be compared with the CPU's bit for bit; what THAT comparison pins is the
// device build (LDS terrain, wave-level control flow, the compiler).  The physics itself is checked by the behavioural
// tests and against an independently written float64 integration of the lander (oracle/lander64.py,
// tests/test_oracle_lander.py::test_independent_float64_lander_envelope).
//
// Box2D itself is third-party, absent from the reference tree and from this image: PARITY WITH BOX2D IS UNPINNED.
// What is restated, from the published Box2D 2.3.0 algorithms (file / function names in the comments below):
//   * bodies with mass, inertia and local centre from b2PolygonShape::ComputeMass (tables: ses_b2_shapes.h);
//   * b2World::Step = Collide (b2CollideEdgeAndPolygon manifolds against the terrain edges, contact ids, warm-start
//     impulse matching, begin / end events) + b2Island::Solve (integrate velocities, b2ContactSolver with friction,
//     two-point block solver and restitution 0, b2RevoluteJoint with limit + motor, warm starting, N velocity
//     iterations, position integration with the translation / rotation caps, up to M position iterations with
//     Box2D's own early exit, island sleep timer);
//   * float32 throughout, IEEE division and sqrt.  Box2D's x86 build rounds every product separately; so does this
//     file, except in the two innermost loops (joint and contact velocity iterations), which use fused multiply-adds.
//   * b2World::SolveTOI -- continuous collision against the terrain: b2Distance / b2SeparationFunction /
//     b2TimeOfImpact (ses_b2_toi.h) per (body polygon, candidate terrain edge) pair and step, the earliest impact rolls
//     its body back, updates its manifolds and sub-steps it alone (b2Island::SolveTOI), see world_solve_toi below.
// Not restated (documented deviations): the broad phase (every terrain edge whose x-range the fattened polygon AABB
// overlaps is a candidate -- same touching set), Box2D's island traversal order (here: joints in definition order, then
// contacts by body, then by manifold slot), sinf / cosf of the C library (here: the build's deterministic sincos), the
// per-contact TOI cache (every pair is re-evaluated after an event), a sub-step for a hull that touches (the touch ends
// the episode; what the sub-step would do is never observed).
//
// The N velocity iterations of Box2D are all run -- or, what has the same result bit for bit, the loop is left once an
// iteration has returned every velocity and every accumulated impulse with the bits it started from: the iteration is a
// deterministic map of exactly those values, so all later iterations are identities.  Two places use that: the lander's main
// loop compares once, after iteration D::VEL_FIXED_POINT_CHECK (ses_lander_env.h has the census; with the light legs on the
// heavy hull the joint rows otherwise converge by about 3 % per iteration and no fixed point comes), and a time-of-impact
// sub-step compares after every iteration (one body against one or two manifolds: ~25 iterations).  Both comparisons are
// bitwise (-0 is not +0).  A build with -DB2_RUN_ALL_ITERATIONS takes neither exit; tests/test_oracle_lander.py holds the
// two builds of the host oracle to identical trajectories.  Quantities that Box2D recomputes in every
// iteration from values that do not change during a step (the inverse of the joint's 3x3 mass matrix) are computed once
// per step and applied as a matrix-vector product: same algorithm, results differ from Box2D's at rounding level.
#pragma once
#include <stdint.h>

#ifndef B2_FN
#error "include through ses_b2_oracle.cpp (host) or csrc/ses_lander.h (device): they define the B2_* macros"
#endif

#include "ses_b2_shapes.h"

namespace b2l {

// ---- b2Settings.h ----
constexpr float PI = 3.14159265359f;
constexpr float LINEAR_SLOP = 0.005f;
constexpr float ANGULAR_SLOP = 2.0f / 180.0f * PI;
constexpr float POLY_RADIUS = 2.0f * LINEAR_SLOP;
constexpr float MAX_LINEAR_CORRECTION = 0.2f;
constexpr float MAX_ANGULAR_CORRECTION = 8.0f / 180.0f * PI;
constexpr float MAX_TRANSLATION = 2.0f;
constexpr float MAX_TRANSLATION_SQ = MAX_TRANSLATION * MAX_TRANSLATION;
constexpr float MAX_ROTATION = 0.5f * PI;
constexpr float MAX_ROTATION_SQ = MAX_ROTATION * MAX_ROTATION;
constexpr float BAUMGARTE = 0.2f;
constexpr float TIME_TO_SLEEP = 0.5f;
constexpr float LINEAR_SLEEP_TOL = 0.01f;
constexpr float ANGULAR_SLEEP_TOL = 2.0f / 180.0f * PI;
constexpr float AABB_EXTENSION = 0.1f;
constexpr float FLT_BIG = 3.402823466e+38f;

B2_FN float b2min(float a, float b) { return a < b ? a : b; }
B2_FN float b2max(float a, float b) { return a > b ? a : b; }
B2_FN float b2clamp(float a, float lo, float hi) { return b2max(lo, b2min(a, hi)); }
B2_FN float b2abs(float a) { return a > 0.0f ? a : -a; }
// clamp to [-lim, lim] in the solver's inner loops: fmin / fmax (one instruction each on the device; same value as
// b2clamp for every non-NaN input)
#ifndef B2_CLAMP_SYM
#define B2_CLAMP_SYM(a, lim) __builtin_fmaxf(-(lim), __builtin_fminf((a), (lim)))
#endif
B2_FN float b2clamp_sym(float a, float lim) { return B2_CLAMP_SYM(a, lim); }   // device: one v_med3_f32 (lim >= 0, no NaN: same value)

struct JointDef {
    int a, b;                        // body indices
    float lax, lay, lbx, lby;        // local anchors
    float lower, upper;              // limits (enableLimit = true), referenceAngle = 0
};

struct Body {
    float cx, cy, a;                 // b2Sweep::c, a
    float vx, vy, w;
};

struct Xf {                          // b2Transform of a body: q = (s, c), p
    float s, c, px, py;
};

constexpr int LIMIT_INACTIVE = 0, LIMIT_LOWER = 1, LIMIT_UPPER = 2, LIMIT_EQUAL = 3;

struct Joint {                       // b2RevoluteJoint, the part that lives across steps
    float ix, iy, iz, im;            // m_impulse, m_motorImpulse (warm starting)
    int state;
    float motor_speed, max_torque;
};

struct JointTmp {                    // InitVelocityConstraints results, alive for one step
    float rax, ray, rbx, rby;
    float n00, n01, n02, n11, n12, n22;   // MINUS the inverse of the symmetric 3x3 m_mass
    float motor_mass, max_impulse;   // max_impulse = dt * maxMotorTorque
    float release_sign;              // +1 at the lower limit, -1 otherwise: the limit lets go when sign * impulse < 0
};

struct JointRare {                   // the part of it that only a releasing limit reads (joint_solve_velocity): kept in
    float ezx, ezy;                  // memory, not in registers, through the iterations -- m_mass.ez.x, .ez.y (the
    float j00, j01, j11;             // limit-release right-hand side) and the inverse of m_mass's upper-left 2x2 block
};

// A pointer the compiler knows nothing about: what is stored through it stays in memory (device build: five registers
// per joint less in the 180-iteration loop, which is what the last in-loop spills of the walker's solver were about).
// development builds (-DSES_PHASE_TIMERS, tools/walker_phases.py) time the phases of a world step; nothing otherwise
#ifndef B2_PHASE_ROWS
#define B2_PHASE_ROWS(mask, iters)
#endif
#ifndef B2_PHASE
#define B2_PHASE(k)
#endif

#ifndef B2_OPAQUE_PTR
#define B2_OPAQUE_PTR(p) asm volatile("" : "+r"(p)::"memory")
#endif

struct Manifold {                    // b2Manifold of (terrain edge `edge`, this body's polygon) + solver temporaries
    int edge;                        // -1: the slot is empty
    int count;                       // pointCount; > 0 = touching
    int type;                        // 0 = e_faceA (the edge is the reference face), 1 = e_faceB
    float lnx, lny, lpx, lpy;        // localNormal, localPoint
    float px[2], py[2];              // points[i].localPoint
    uint32_t id[2];                  // b2ContactID::key
    float ni[2], ti[2];              // normalImpulse, tangentImpulse
};

struct ContactTmp {                  // b2ContactVelocityConstraint, alive for one step
    int vcount;                      // pointCount after the block solver's condition test
    float nx, ny;
    float rbx[2], rby[2], nmass[2], tmass[2];
    float k11, k12, k22, b11, b12, b21, b22;   // K and normalMass (= K^-1): b11 = ex.x, b12 = ey.x, b21 = ex.y, b22 = ey.y
};

// World description: D provides
//   static constexpr int NB, NJ, NSLOT (power of two), FIRST_SOLVED (bodies >= this have their contacts solved; body 0 =
//   hull: touching the terrain ends the episode), VEL_ITERS, POS_ITERS; static constexpr bool PACK_MANIFOLDS, CONTINUOUS
//   static const Poly *poly(); static const BodyDef *body(); static const JointDef *joint();
template <class D>
struct World {
    Body body[D::NB];
    Xf xf[D::NB];                    // transforms at the start of the current step (Collide / solver initialisation)
    Joint joint[D::NJ];
    Manifold mf[D::NB - D::FIRST_SOLVED][D::NSLOT];   // of the bodies whose contacts are solved (index b - FIRST_SOLVED)
    float sleep_time[D::NB];
    bool ground_contact[D::NB];
    bool game_over, awake;
    float fx, fy;                    // b2Body::m_force of body 0 (ApplyForceToCenter), cleared after a step
};

B2_FN void xf_of(const Body &b, const BodyDef &d, Xf &x)
{
    float s, c;
    B2_SINCOS(b.a, s, c);
    x.s = s; x.c = c;
    x.px = b.cx - (c * d.lcx - s * d.lcy);
    x.py = b.cy - (s * d.lcx + c * d.lcy);
}

// ------------------------------------------------------------------------------------------------------------------
// b2CollideEdgeAndPolygon (b2CollideEdge.cpp, b2EPCollider::Collide) for an edge without adjacent vertices, the edge
// in world coordinates (the terrain body sits at the origin with angle 0: its transform is the identity, so
// "frame A" = world and m_xf = xfB).
struct ClipVertex {
    float x, y;
    uint32_t id;
};
B2_FN uint32_t contact_id(int indexA, int indexB, int typeA, int typeB)
{
    return (uint32_t)indexA | ((uint32_t)indexB << 8) | ((uint32_t)typeA << 16) | ((uint32_t)typeB << 24);
}
constexpr int CF_VERTEX = 0, CF_FACE = 1;

B2_FN int clip_segment_to_line(ClipVertex (&out)[2], const ClipVertex (&in)[2], float nx, float ny, float offset, int vertexIndexA)
{
    const float d0 = (nx * in[0].x + ny * in[0].y) - offset;
    const float d1 = (nx * in[1].x + ny * in[1].y) - offset;
    const bool in0 = d0 <= 0.0f, in1 = d1 <= 0.0f;
    out[0] = in0 ? in[0] : in[1];                  // vOut[numOut++] = vIn[0] / vIn[1] for the points behind the plane
    out[1] = in[1];
    int num = (in0 ? 1 : 0) + (in1 ? 1 : 0);
    if (d0 * d1 < 0.0f) {                          // exactly one point is behind the plane: num is 1 here
        const float interp = d0 / (d0 - d1);
        out[1].x = in[0].x + interp * (in[1].x - in[0].x);
        out[1].y = in[0].y + interp * (in[1].y - in[0].y);
        out[1].id = contact_id(vertexIndexA, (int)((in[0].id >> 8) & 0xffu), CF_VERTEX, CF_FACE);
        num = 2;
    }
    return num;
}

template <int MAXV>
B2_FN float pick(const float (&arr)[MAXV], int n, int idx)
{
    float r = arr[0];
    B2_UNROLL
    for (int i = 1; i < MAXV; ++i) r = (i < n && i == idx) ? arr[i] : r;
    return r;
}

// polygon in world coordinates: vertices (wx, wy), normals (wnx, wny), centroid (ccx, ccy); the body transform xf and
// the local polygon are needed for the manifold's local points.  Fills the geometric part of `m` (count, type,
// local normal / point, points, ids); impulses are left to the caller.
B2_FN void collide_edge_polygon(const Poly &P, const Xf &xf, const float (&wx)[6], const float (&wy)[6],
                                const float (&wnx)[6], const float (&wny)[6], float ccx, float ccy,
                                float v1x, float v1y, float v2x, float v2y, Manifold &m)
{
    m.count = 0;
    const int n = P.n;
    float e1x = v2x - v1x, e1y = v2y - v1y;
    {
        const float len = B2_SQRT(e1x * e1x + e1y * e1y);       // b2Vec2::Normalize
        if (!(len < 1.1920928955078125e-7f)) {
            const float inv = 1.0f / len;
            e1x *= inv; e1y *= inv;
        }
    }
    const float n1x = e1y, n1y = -e1x;                            // m_normal1
    const float offset1 = n1x * (ccx - v1x) + n1y * (ccy - v1y);
    const bool front = offset1 >= 0.0f;
    const float mnx = front ? n1x : -n1x, mny = front ? n1y : -n1y;           // m_normal
    const float limx = front ? -n1x : n1x, limy = front ? -n1y : n1y;          // m_lowerLimit = m_upperLimit
    const float radius = 2.0f * POLY_RADIUS;

    // ComputeEdgeSeparation
    float edge_sep = FLT_BIG;
    B2_UNROLL
    for (int i = 0; i < 6; ++i) {
        if (i < n) {
            const float s = mnx * (wx[i] - v1x) + mny * (wy[i] - v1y);
            if (s < edge_sep) edge_sep = s;
        }
    }
    if (edge_sep > radius) return;

    // ComputePolygonSeparation
    int poly_index = -1;
    float poly_sep = -FLT_BIG;
    bool poly_early = false;
    B2_UNROLL
    for (int i = 0; i < 6; ++i) {
        if (i < n && !poly_early) {
            const float nx = -wnx[i], ny = -wny[i];
            const float s1 = nx * (wx[i] - v1x) + ny * (wy[i] - v1y);
            const float s2 = nx * (wx[i] - v2x) + ny * (wy[i] - v2y);
            const float s = b2min(s1, s2);
            if (s > radius) {
                poly_index = i; poly_sep = s; poly_early = true;
            } else {
                // adjacency: Box2D picks the upper or the lower limit by the side of `perp` the normal lies on; for an
                // isolated edge the two limits coincide
                const bool skip = ((nx - limx) * mnx + (ny - limy) * mny) < -ANGULAR_SLOP;
                if (!skip && s > poly_sep) { poly_index = i; poly_sep = s; }
            }
        }
    }
    if (poly_index >= 0 && poly_sep > radius) return;

    const float k_relativeTol = 0.98f, k_absoluteTol = 0.001f;
    const bool primary_edge = poly_index < 0 || !(poly_sep > k_relativeTol * edge_sep + k_absoluteTol);

    ClipVertex ie[2];
    int rf_i1, rf_i2;
    float rf_v1x, rf_v1y, rf_v2x, rf_v2y, rf_nx, rf_ny;
    if (primary_edge) {
        m.type = 0;
        int best = 0;
        float best_value = mnx * wnx[0] + mny * wny[0];
        B2_UNROLL
        for (int i = 1; i < 6; ++i) {
            if (i < n) {
                const float value = mnx * wnx[i] + mny * wny[i];
                if (value < best_value) { best_value = value; best = i; }
            }
        }
        const int i1 = best, i2 = i1 + 1 < n ? i1 + 1 : 0;
        ie[0].x = pick(wx, n, i1); ie[0].y = pick(wy, n, i1); ie[0].id = contact_id(0, i1, CF_FACE, CF_VERTEX);
        ie[1].x = pick(wx, n, i2); ie[1].y = pick(wy, n, i2); ie[1].id = contact_id(0, i2, CF_FACE, CF_VERTEX);
        if (front) {
            rf_i1 = 0; rf_i2 = 1; rf_v1x = v1x; rf_v1y = v1y; rf_v2x = v2x; rf_v2y = v2y; rf_nx = n1x; rf_ny = n1y;
        } else {
            rf_i1 = 1; rf_i2 = 0; rf_v1x = v2x; rf_v1y = v2y; rf_v2x = v1x; rf_v2y = v1y; rf_nx = -n1x; rf_ny = -n1y;
        }
    } else {
        m.type = 1;
        ie[0].x = v1x; ie[0].y = v1y; ie[0].id = contact_id(0, poly_index, CF_VERTEX, CF_FACE);
        ie[1].x = v2x; ie[1].y = v2y; ie[1].id = contact_id(0, poly_index, CF_VERTEX, CF_FACE);
        rf_i1 = poly_index; rf_i2 = rf_i1 + 1 < n ? rf_i1 + 1 : 0;
        rf_v1x = pick(wx, n, rf_i1); rf_v1y = pick(wy, n, rf_i1);
        rf_v2x = pick(wx, n, rf_i2); rf_v2y = pick(wy, n, rf_i2);
        rf_nx = pick(wnx, n, rf_i1); rf_ny = pick(wny, n, rf_i1);
    }
    const float sn1x = rf_ny, sn1y = -rf_nx, sn2x = -sn1x, sn2y = -sn1y;
    const float so1 = sn1x * rf_v1x + sn1y * rf_v1y;
    const float so2 = sn2x * rf_v2x + sn2y * rf_v2y;
    ClipVertex c1[2], c2[2];
    if (clip_segment_to_line(c1, ie, sn1x, sn1y, so1, rf_i1) < 2) return;
    if (clip_segment_to_line(c2, c1, sn2x, sn2y, so2, rf_i2) < 2) return;

    if (primary_edge) {
        m.lnx = rf_nx; m.lny = rf_ny; m.lpx = rf_v1x; m.lpy = rf_v1y;
    } else {
        m.lnx = pick(P.nx, n, rf_i1); m.lny = pick(P.ny, n, rf_i1);
        m.lpx = pick(P.vx, n, rf_i1); m.lpy = pick(P.vy, n, rf_i1);
    }
    int count = 0;
    B2_UNROLL
    for (int i = 0; i < 2; ++i) {
        const float sep = rf_nx * (c2[i].x - rf_v1x) + rf_ny * (c2[i].y - rf_v1y);
        if (sep <= radius) {
            float lx, ly;
            uint32_t id;
            if (primary_edge) {                                   // b2MulT(m_xf, v)
                const float dx = c2[i].x - xf.px, dy = c2[i].y - xf.py;
                lx = xf.c * dx + xf.s * dy;
                ly = -xf.s * dx + xf.c * dy;
                id = c2[i].id;
            } else {
                lx = c2[i].x; ly = c2[i].y;
                const uint32_t k = c2[i].id;                      // swap the A and B features
                id = ((k >> 8) & 0xffu) | ((k & 0xffu) << 8) | (((k >> 24) & 0xffu) << 16) | (((k >> 16) & 0xffu) << 24);
            }
            if (count == 0) { m.px[0] = lx; m.py[0] = ly; m.id[0] = id; }
            else { m.px[1] = lx; m.py[1] = ly; m.id[1] = id; }
            ++count;
        }
    }
    m.count = count;
}

// ------------------------------------------------------------------------------------------------------------------
// b2World::Step, first half: b2ContactManager::Collide over the (terrain edge, body polygon) pairs.
// T: terrain with  int n_edges() const;  int index_of(float x) const  (edge containing x, unclamped);
//                  void edge(int k, float &x1, float &y1, float &x2, float &y2) const.
template <class D, class T>
B2_FN void collide_body(World<D> &w, const T &terr, const int b)
{
    {
        const Poly &P = D::poly()[b];
        const BodyDef &bd = D::body()[b];
        xf_of(w.body[b], bd, w.xf[b]);
        const Xf &xf = w.xf[b];
        float wx[6], wy[6], wnx[6], wny[6];
        float xmin = FLT_BIG, xmax = -FLT_BIG, ymin = FLT_BIG;
        B2_UNROLL
        for (int i = 0; i < 6; ++i) {
            if (i < P.n) {
                wx[i] = (xf.c * P.vx[i] - xf.s * P.vy[i]) + xf.px;
                wy[i] = (xf.s * P.vx[i] + xf.c * P.vy[i]) + xf.py;
                wnx[i] = xf.c * P.nx[i] - xf.s * P.ny[i];
                wny[i] = xf.s * P.nx[i] + xf.c * P.ny[i];
                xmin = b2min(xmin, wx[i]); xmax = b2max(xmax, wx[i]); ymin = b2min(ymin, wy[i]);
            } else {
                wx[i] = 0.0f; wy[i] = 0.0f; wnx[i] = 0.0f; wny[i] = 0.0f;
            }
        }
        const float ccx = (xf.c * P.cx - xf.s * P.cy) + xf.px, ccy = (xf.s * P.cx + xf.c * P.cy) + xf.py;
        int k_lo = terr.index_of(xmin - AABB_EXTENSION), k_hi = terr.index_of(xmax + AABB_EXTENSION);
        k_lo = k_lo < 0 ? 0 : k_lo;
        k_hi = k_hi > terr.n_edges() - 1 ? terr.n_edges() - 1 : k_hi;
        if (b < D::FIRST_SOLVED) {
            // the hull: touching the terrain ends the episode (the envs' contact listeners), nothing is solved or kept
            for (int k = k_lo; k <= k_hi; ++k) {
                float x1, y1, x2, y2;
                terr.edge(k, x1, y1, x2, y2);
                if (!(ymin - AABB_EXTENSION > b2max(y1, y2))) {
                    Manifold tmp;
                    collide_edge_polygon(P, xf, wx, wy, wnx, wny, ccx, ccy, x1, y1, x2, y2, tmp);
                    if (tmp.count > 0) w.game_over = true;
                }
            }
            return;
        }
        k_hi = k_hi > k_lo + D::NSLOT - 1 ? k_lo + D::NSLOT - 1 : k_hi;
        B2_UNROLL
        for (int s = 0; s < D::NSLOT; ++s) {
            Manifold &m = w.mf[b - D::FIRST_SOLVED][s];
            const int k = k_lo + ((s - k_lo) & (D::NSLOT - 1));       // the candidate edge that maps to this slot
            const bool candidate = k <= k_hi && k_lo <= k_hi;
            const bool was_touching = m.edge >= 0 && m.count > 0;
            const bool same_edge = candidate && m.edge == k;
            Manifold old = m;
            m.count = 0;
            m.edge = candidate ? k : -1;
            if (candidate) {
                float x1, y1, x2, y2;
                terr.edge(k, x1, y1, x2, y2);
                // (the manifold function returns no points for anything farther than 2 * polygonRadius; this test
                //  only skips its evaluation for a polygon that is clearly above the edge)
                if (!(ymin - AABB_EXTENSION > b2max(y1, y2)))
                    collide_edge_polygon(P, xf, wx, wy, wnx, wny, ccx, ccy, x1, y1, x2, y2, m);
            }
            // b2Contact::Update: impulses of the points whose ids persist
            B2_UNROLL
            for (int i = 0; i < 2; ++i) {
                m.ni[i] = 0.0f; m.ti[i] = 0.0f;
                if (same_edge && i < m.count) {
                    B2_UNROLL
                    for (int j = 0; j < 2; ++j) {
                        if (j < old.count && old.id[j] == m.id[i]) { m.ni[i] = old.ni[j]; m.ti[i] = old.ti[j]; break; }
                    }
                }
            }
            const bool still = same_edge && was_touching;
            if (was_touching && !(still && m.count > 0)) w.ground_contact[b] = false;          // EndContact
            if (m.count > 0 && !still) w.ground_contact[b] = true;                             // BeginContact
        }
    }
}

template <class D, class T>
B2_FN void collide(World<D> &w, const T &terr)
{
    B2_UNROLL
    for (int b = 0; b < D::NB; ++b) collide_body(w, terr, b);
}

// ------------------------------------------------------------------------------------------------------------------
// b2RevoluteJoint
template <class D>
B2_FN void joint_init(Body (&body)[D::NB], Joint (&joint)[D::NJ], const Xf (&xf)[D::NB], int j, JointTmp &t, JointRare *rare, float dt)
{
    const JointDef &jd = D::joint()[j];
    Joint &J = joint[j];
    const BodyDef &da = D::body()[jd.a], &db = D::body()[jd.b];
    const Xf &qa = xf[jd.a], &qb = xf[jd.b];
    Body &A = body[jd.a], &B = body[jd.b];
    {
        const float ax = jd.lax - da.lcx, ay = jd.lay - da.lcy, bx = jd.lbx - db.lcx, by = jd.lby - db.lcy;
        t.rax = qa.c * ax - qa.s * ay; t.ray = qa.s * ax + qa.c * ay;
        t.rbx = qb.c * bx - qb.s * by; t.rby = qb.s * bx + qb.c * by;
    }
    const float mA = da.inv_mass, mB = db.inv_mass, iA = da.inv_i, iB = db.inv_i;
    // m_mass (symmetric): ex = (exx, eyx, ezx), ey = (eyx, eyy, ezy), ez = (ezx, ezy, ezz)
    const float exx = mA + mB + t.ray * t.ray * iA + t.rby * t.rby * iB;
    const float eyx = -t.ray * t.rax * iA - t.rby * t.rbx * iB;
    const float ezx = -t.ray * iA - t.rby * iB;
    const float eyy = mA + mB + t.rax * t.rax * iA + t.rbx * t.rbx * iB;
    const float ezy = t.rax * iA + t.rbx * iB;
    const float ezz = iA + iB;
    rare[j].ezx = ezx; rare[j].ezy = ezy;
    float j00, j01, j11;
    {   // inverse by cofactors (b2Mat33::Solve33 divides by the same determinant)
        const float c00 = eyy * ezz - ezy * ezy, c01 = ezy * ezx - eyx * ezz, c02 = eyx * ezy - eyy * ezx;
        float det = exx * c00 + eyx * c01 + ezx * c02;
        if (det != 0.0f) det = 1.0f / det;
        const float c11 = exx * ezz - ezx * ezx, c12 = eyx * ezx - exx * ezy, c22 = exx * eyy - eyx * eyx;
        t.n00 = -(det * c00); t.n01 = -(det * c01); t.n02 = -(det * c02);
        t.n11 = -(det * c11); t.n12 = -(det * c12); t.n22 = -(det * c22);
        float d2 = exx * eyy - eyx * eyx;                    // b2Mat33::Solve22
        if (d2 != 0.0f) d2 = 1.0f / d2;
        j00 = d2 * eyy; j01 = -d2 * eyx; j11 = d2 * exx;
        rare[j].j00 = j00; rare[j].j01 = j01; rare[j].j11 = j11;
    }
    t.motor_mass = iA + iB;
    if (t.motor_mass > 0.0f) t.motor_mass = 1.0f / t.motor_mass;
    t.max_impulse = dt * J.max_torque;
    {
        const float angle = B.a - A.a;                       // referenceAngle = 0
        if (b2abs(jd.upper - jd.lower) < 2.0f * ANGULAR_SLOP) {
            J.state = LIMIT_EQUAL;
        } else if (angle <= jd.lower) {
            if (J.state != LIMIT_LOWER) J.iz = 0.0f;
            J.state = LIMIT_LOWER;
        } else if (angle >= jd.upper) {
            if (J.state != LIMIT_UPPER) J.iz = 0.0f;
            J.state = LIMIT_UPPER;
        } else {
            J.state = LIMIT_INACTIVE;
            J.iz = 0.0f;
        }
    }
    t.release_sign = J.state == LIMIT_LOWER ? 1.0f : -1.0f;
    if (J.state == LIMIT_INACTIVE) {      // no limit row this step: the point rows alone (2x2 block), written in the
        t.n00 = -j00; t.n01 = -j01; t.n11 = -j11;              // same form so that the iteration below has one path
        t.n02 = 0.0f; t.n12 = 0.0f; t.n22 = 0.0f;
    }
    // warm start (dtRatio = 1 for a constant time step: 50.0f * 0.02f rounds to 1.0f)
    const float Px = J.ix, Py = J.iy;
    A.vx -= mA * Px; A.vy -= mA * Py;
    A.w -= iA * ((t.rax * Py - t.ray * Px) + J.im + J.iz);
    B.vx += mB * Px; B.vy += mB * Py;
    B.w += iB * ((t.rbx * Py - t.rby * Px) + J.im + J.iz);
}

// One velocity iteration of the joint: motor row, then the point rows together with the limit row (b2RevoluteJoint::
// SolveVelocityConstraints).  This is the innermost loop of the world (180 x NJ per step), so it is written with fused
// multiply-adds and the pre-negated inverse mass matrix; Box2D's x86 build rounds every product separately.
template <class D>
B2_FN void joint_solve_velocity(Body (&body)[D::NB], Joint (&joint)[D::NJ], int j, const JointTmp &t, const JointRare *rare)
{
    const JointDef &jd = D::joint()[j];
    Joint &J = joint[j];
    const BodyDef &da = D::body()[jd.a], &db = D::body()[jd.b];
    Body &A = body[jd.a], &B = body[jd.b];
    const float mA = da.inv_mass, mB = db.inv_mass, iA = da.inv_i, iB = db.inv_i;
    const bool equal_limits = b2abs(jd.upper - jd.lower) < 2.0f * ANGULAR_SLOP;   // a constant of the joint
    if (!equal_limits) {                                      // motor (enableMotor = true)
        const float Cdot = (B.w - A.w) - J.motor_speed;
        const float old = J.im;
        J.im = b2clamp_sym(__builtin_fmaf(-t.motor_mass, Cdot, old), t.max_impulse);
        const float impulse = J.im - old;
        A.w = __builtin_fmaf(-iA, impulse, A.w);
        B.w = __builtin_fmaf(iB, impulse, B.w);
    }
    const float c1x = __builtin_fmaf(-B.w, t.rby, B.vx) - __builtin_fmaf(-A.w, t.ray, A.vx);
    const float c1y = __builtin_fmaf(B.w, t.rbx, B.vy) - __builtin_fmaf(A.w, t.rax, A.vy);
    const float c2 = B.w - A.w;
    // impulse = -M^-1 Cdot: 3x3 with an active limit, the 2x2 block padded with zeros without (joint_init)
    float ix = __builtin_fmaf(t.n00, c1x, __builtin_fmaf(t.n01, c1y, t.n02 * c2));
    float iy = __builtin_fmaf(t.n01, c1x, __builtin_fmaf(t.n11, c1y, t.n12 * c2));
    float iz = __builtin_fmaf(t.n02, c1x, __builtin_fmaf(t.n12, c1y, t.n22 * c2));
    if (!equal_limits) {
        const float new_impulse = J.iz + iz;                  // (inactive limit: J.iz = iz = 0, never released)
        // lower limit: new_impulse < 0, upper: new_impulse > 0 -- one multiply by +-1 (exact) and one compare
        const bool release = new_impulse * t.release_sign < 0.0f;
        if (release) {                                         // the limit lets go: solve the point rows alone
            B2_RARE_PATH;                                      // (rare: keep it a branch, not a select over both results)
            const JointRare q = rare[j];
            const float rx = __builtin_fmaf(J.iz, q.ezx, -c1x), ry = __builtin_fmaf(J.iz, q.ezy, -c1y);
            ix = __builtin_fmaf(q.j00, rx, q.j01 * ry);
            iy = __builtin_fmaf(q.j01, rx, q.j11 * ry);
            iz = -J.iz;
        }
    }
    J.ix += ix; J.iy += iy; J.iz += iz;
    A.vx = __builtin_fmaf(-mA, ix, A.vx); A.vy = __builtin_fmaf(-mA, iy, A.vy);
    A.w = __builtin_fmaf(-iA, __builtin_fmaf(t.rax, iy, -(t.ray * ix)) + iz, A.w);
    B.vx = __builtin_fmaf(mB, ix, B.vx); B.vy = __builtin_fmaf(mB, iy, B.vy);
    B.w = __builtin_fmaf(iB, __builtin_fmaf(t.rbx, iy, -(t.rby * ix)) + iz, B.w);
}

template <class D>
B2_FN bool joint_solve_position(Body (&body)[D::NB], const Joint (&joint)[D::NJ], int j)
{
    const JointDef &jd = D::joint()[j];
    const Joint &J = joint[j];
    const BodyDef &da = D::body()[jd.a], &db = D::body()[jd.b];
    Body &A = body[jd.a], &B = body[jd.b];
    const float mA = da.inv_mass, mB = db.inv_mass, iA = da.inv_i, iB = db.inv_i;
    float motor_mass = iA + iB;
    if (motor_mass > 0.0f) motor_mass = 1.0f / motor_mass;
    float angular_error = 0.0f;
    if (J.state != LIMIT_INACTIVE) {
        const float angle = B.a - A.a;
        float limit_impulse = 0.0f;
        if (J.state == LIMIT_EQUAL) {
            const float C = b2clamp(angle - jd.lower, -MAX_ANGULAR_CORRECTION, MAX_ANGULAR_CORRECTION);
            limit_impulse = -motor_mass * C;
            angular_error = b2abs(C);
        } else if (J.state == LIMIT_LOWER) {
            float C = angle - jd.lower;
            angular_error = -C;
            C = b2clamp(C + ANGULAR_SLOP, -MAX_ANGULAR_CORRECTION, 0.0f);
            limit_impulse = -motor_mass * C;
        } else {
            float C = angle - jd.upper;
            angular_error = C;
            C = b2clamp(C - ANGULAR_SLOP, 0.0f, MAX_ANGULAR_CORRECTION);
            limit_impulse = -motor_mass * C;
        }
        A.a -= iA * limit_impulse;
        B.a += iB * limit_impulse;
    }
    float sa, ca, sb, cb;
    B2_SINCOS(A.a, sa, ca);
    B2_SINCOS(B.a, sb, cb);
    const float ax = jd.lax - da.lcx, ay = jd.lay - da.lcy, bx = jd.lbx - db.lcx, by = jd.lby - db.lcy;
    const float rax = ca * ax - sa * ay, ray = sa * ax + ca * ay;
    const float rbx = cb * bx - sb * by, rby = sb * bx + cb * by;
    const float Cx = B.cx + rbx - A.cx - rax, Cy = B.cy + rby - A.cy - ray;
    const float position_error = B2_SQRT(Cx * Cx + Cy * Cy);
    const float k11 = mA + mB + iA * ray * ray + iB * rby * rby;
    const float k12 = -iA * rax * ray - iB * rbx * rby;
    const float k22 = mA + mB + iA * rax * rax + iB * rbx * rbx;
    float det = k11 * k22 - k12 * k12;
    if (det != 0.0f) det = 1.0f / det;
    const float ix = -(det * (k22 * Cx - k12 * Cy)), iy = -(det * (k11 * Cy - k12 * Cx));
    A.cx -= mA * ix; A.cy -= mA * iy;
    A.a -= iA * (rax * iy - ray * ix);
    B.cx += mB * ix; B.cy += mB * iy;
    B.a += iB * (rbx * iy - rby * ix);
    return position_error <= LINEAR_SLOP && angular_error <= ANGULAR_SLOP;
}

// ------------------------------------------------------------------------------------------------------------------
// b2ContactSolver for (static terrain at the identity transform = body A, dynamic polygon body = body B): every
// A-side term is a product with invMassA = invIA = 0 or a difference with vA = wA = 0 and drops out exactly.
B2_FN void contact_init(const Manifold &m, ContactTmp &t, const Body &B, const BodyDef &bd, const Xf &xf)
{
    t.vcount = m.count;
    if (m.count == 0) return;
    const float mB = bd.inv_mass, iB = bd.inv_i;
    float nx, ny;
    float ptx[2], pty[2];
    if (m.type == 0) {                                        // b2WorldManifold::Initialize, e_faceA
        nx = m.lnx; ny = m.lny;
        B2_UNROLL
        for (int i = 0; i < 2; ++i) {
            const float cpx = (xf.c * m.px[i] - xf.s * m.py[i]) + xf.px, cpy = (xf.s * m.px[i] + xf.c * m.py[i]) + xf.py;
            const float d = POLY_RADIUS - ((cpx - m.lpx) * nx + (cpy - m.lpy) * ny);
            const float cAx = cpx + d * nx, cAy = cpy + d * ny;
            const float cBx = cpx - POLY_RADIUS * nx, cBy = cpy - POLY_RADIUS * ny;
            ptx[i] = 0.5f * (cAx + cBx); pty[i] = 0.5f * (cAy + cBy);
        }
    } else {                                                  // e_faceB
        nx = xf.c * m.lnx - xf.s * m.lny; ny = xf.s * m.lnx + xf.c * m.lny;
        const float ppx = (xf.c * m.lpx - xf.s * m.lpy) + xf.px, ppy = (xf.s * m.lpx + xf.c * m.lpy) + xf.py;
        B2_UNROLL
        for (int i = 0; i < 2; ++i) {
            const float cpx = m.px[i], cpy = m.py[i];
            const float d = POLY_RADIUS - ((cpx - ppx) * nx + (cpy - ppy) * ny);
            const float cBx = cpx + d * nx, cBy = cpy + d * ny;
            const float cAx = cpx - POLY_RADIUS * nx, cAy = cpy - POLY_RADIUS * ny;
            ptx[i] = 0.5f * (cAx + cBx); pty[i] = 0.5f * (cAy + cBy);
        }
        nx = -nx; ny = -ny;
    }
    t.nx = nx; t.ny = ny;
    const float tx = ny, ty = -nx;                            // b2Cross(normal, 1.0f)
    float rn[2];
    B2_UNROLL
    for (int i = 0; i < 2; ++i) {
        t.rbx[i] = ptx[i] - B.cx; t.rby[i] = pty[i] - B.cy;
        rn[i] = t.rbx[i] * ny - t.rby[i] * nx;
        const float kn = mB + iB * rn[i] * rn[i];
        t.nmass[i] = kn > 0.0f ? 1.0f / kn : 0.0f;
        const float rt = t.rbx[i] * ty - t.rby[i] * tx;
        const float kt = mB + iB * rt * rt;
        t.tmass[i] = kt > 0.0f ? 1.0f / kt : 0.0f;
        // restitution 0: velocityBias = 0
    }
    t.k11 = 0.0f; t.k12 = 0.0f; t.k22 = 0.0f; t.b11 = 0.0f; t.b12 = 0.0f; t.b21 = 0.0f; t.b22 = 0.0f;
    if (m.count == 2) {
        const float k11 = mB + iB * rn[0] * rn[0], k22 = mB + iB * rn[1] * rn[1], k12 = mB + iB * rn[0] * rn[1];
        if (k11 * k11 < 1000.0f * (k11 * k22 - k12 * k12)) {
            t.k11 = k11; t.k12 = k12; t.k22 = k22;
            float det = k11 * k22 - k12 * k12;                // b2Mat22::GetInverse
            if (det != 0.0f) det = 1.0f / det;
            t.b11 = det * k22; t.b12 = -det * k12; t.b21 = -det * k12; t.b22 = det * k11;
        } else {
            t.vcount = 1;
        }
    }
}

B2_FN void contact_warm_start(const Manifold &m, const ContactTmp &t, Body &B, const BodyDef &bd)
{
    const float tx = t.ny, ty = -t.nx;
    B2_UNROLL
    for (int i = 0; i < 2; ++i) {
        if (i < t.vcount) {
            const float Px = m.ni[i] * t.nx + m.ti[i] * tx, Py = m.ni[i] * t.ny + m.ti[i] * ty;
            B.w += bd.inv_i * (t.rbx[i] * Py - t.rby[i] * Px);
            B.vx += bd.inv_mass * Px; B.vy += bd.inv_mass * Py;
        }
    }
}

B2_FN void contact_solve_velocity(Manifold &m, const ContactTmp &t, Body &B, const BodyDef &bd)
{
    if (t.vcount == 0) return;
    const float mB = bd.inv_mass, iB = bd.inv_i, friction = bd.friction;
    const float nx = t.nx, ny = t.ny, tx = ny, ty = -nx;
    B2_UNROLL
    for (int i = 0; i < 2; ++i) {                              // friction first
        if (i < t.vcount) {
            const float dvx = __builtin_fmaf(-B.w, t.rby[i], B.vx), dvy = __builtin_fmaf(B.w, t.rbx[i], B.vy);
            const float vt = __builtin_fmaf(dvx, tx, dvy * ty);
            const float max_friction = friction * m.ni[i];
            const float new_impulse = b2clamp_sym(__builtin_fmaf(-t.tmass[i], vt, m.ti[i]), max_friction);
            const float lambda = new_impulse - m.ti[i];
            m.ti[i] = new_impulse;
            const float Px = lambda * tx, Py = lambda * ty;
            B.vx = __builtin_fmaf(mB, Px, B.vx); B.vy = __builtin_fmaf(mB, Py, B.vy);
            B.w = __builtin_fmaf(iB, __builtin_fmaf(t.rbx[i], Py, -(t.rby[i] * Px)), B.w);
        }
    }
    if (t.vcount == 1) {
        const float dvx = __builtin_fmaf(-B.w, t.rby[0], B.vx), dvy = __builtin_fmaf(B.w, t.rbx[0], B.vy);
        const float vn = __builtin_fmaf(dvx, nx, dvy * ny);
        const float new_impulse = b2max(__builtin_fmaf(-t.nmass[0], vn, m.ni[0]), 0.0f);
        const float lambda = new_impulse - m.ni[0];
        m.ni[0] = new_impulse;
        const float Px = lambda * nx, Py = lambda * ny;
        B.vx = __builtin_fmaf(mB, Px, B.vx); B.vy = __builtin_fmaf(mB, Py, B.vy);
        B.w = __builtin_fmaf(iB, __builtin_fmaf(t.rbx[0], Py, -(t.rby[0] * Px)), B.w);
    } else {                                                   // block solver (b2ContactSolver.cpp, the four cases)
        const float ax = m.ni[0], ay = m.ni[1];
        const float dv1x = __builtin_fmaf(-B.w, t.rby[0], B.vx), dv1y = __builtin_fmaf(B.w, t.rbx[0], B.vy);
        const float dv2x = __builtin_fmaf(-B.w, t.rby[1], B.vx), dv2y = __builtin_fmaf(B.w, t.rbx[1], B.vy);
        float bx = __builtin_fmaf(dv1x, nx, dv1y * ny), by = __builtin_fmaf(dv2x, nx, dv2y * ny);
        bx -= __builtin_fmaf(t.k11, ax, t.k12 * ay);
        by -= __builtin_fmaf(t.k12, ax, t.k22 * ay);
        float xx = -__builtin_fmaf(t.b11, bx, t.b12 * by), xy = -__builtin_fmaf(t.b21, bx, t.b22 * by);      // case 1
        bool solved = xx >= 0.0f && xy >= 0.0f;
        if (!solved) {                                         // case 2
            xx = -t.nmass[0] * bx; xy = 0.0f;
            const float vn2 = __builtin_fmaf(t.k12, xx, by);
            solved = xx >= 0.0f && vn2 >= 0.0f;
        }
        if (!solved) {                                         // case 3
            xx = 0.0f; xy = -t.nmass[1] * by;
            const float vn1 = __builtin_fmaf(t.k12, xy, bx);
            solved = xy >= 0.0f && vn1 >= 0.0f;
        }
        if (!solved) {                                         // case 4
            xx = 0.0f; xy = 0.0f;
            solved = bx >= 0.0f && by >= 0.0f;
        }
        if (solved) {
            const float dx = xx - ax, dy = xy - ay;
            const float P1x = dx * nx, P1y = dx * ny, P2x = dy * nx, P2y = dy * ny;
            B.vx = __builtin_fmaf(mB, P1x + P2x, B.vx); B.vy = __builtin_fmaf(mB, P1y + P2y, B.vy);
            B.w = __builtin_fmaf(iB, __builtin_fmaf(t.rbx[0], P1y, -(t.rby[0] * P1x)) + __builtin_fmaf(t.rbx[1], P2y, -(t.rby[1] * P2x)), B.w);
            m.ni[0] = xx; m.ni[1] = xy;
        }
    }
}

// b2ContactSolver::SolvePositionConstraints for one manifold; returns its minimum separation
B2_FN float contact_solve_position(const Manifold &m, Body &B, const BodyDef &bd, const float baumgarte = BAUMGARTE)
{
    float min_separation = 0.0f;
    const float mB = bd.inv_mass, iB = bd.inv_i;
    B2_UNROLL
    for (int i = 0; i < 2; ++i) {
        if (i < m.count) {
            float s, c;
            B2_SINCOS(B.a, s, c);
            const float px = B.cx - (c * bd.lcx - s * bd.lcy), py = B.cy - (s * bd.lcx + c * bd.lcy);
            float nx, ny, ptx, pty, separation;
            if (m.type == 0) {                                // b2PositionSolverManifold, e_faceA
                nx = m.lnx; ny = m.lny;
                const float cpx = (c * m.px[i] - s * m.py[i]) + px, cpy = (s * m.px[i] + c * m.py[i]) + py;
                separation = ((cpx - m.lpx) * nx + (cpy - m.lpy) * ny) - POLY_RADIUS - POLY_RADIUS;
                ptx = cpx; pty = cpy;
            } else {
                nx = c * m.lnx - s * m.lny; ny = s * m.lnx + c * m.lny;
                const float ppx = (c * m.lpx - s * m.lpy) + px, ppy = (s * m.lpx + c * m.lpy) + py;
                const float cpx = m.px[i], cpy = m.py[i];
                separation = ((cpx - ppx) * nx + (cpy - ppy) * ny) - POLY_RADIUS - POLY_RADIUS;
                ptx = cpx; pty = cpy;
                nx = -nx; ny = -ny;
            }
            const float rbx = ptx - B.cx, rby = pty - B.cy;
            min_separation = b2min(min_separation, separation);
            const float C = b2clamp(baumgarte * (separation + LINEAR_SLOP), -MAX_LINEAR_CORRECTION, 0.0f);
            const float rn = rbx * ny - rby * nx;
            const float K = mB + iB * rn * rn;
            const float impulse = K > 0.0f ? -C / K : 0.0f;
            const float Px = impulse * nx, Py = impulse * ny;
            B.cx += mB * Px; B.cy += mB * Py;
            B.a += iB * (rbx * Py - rby * Px);
        }
    }
    return min_separation;
}

}  // namespace b2l
#include "ses_b2_toi.h"
namespace b2l {

// ------------------------------------------------------------------------------------------------------------------
// b2World::Step(dt, D::VEL_ITERS, D::POS_ITERS)
//
// b2Island::Solve after the velocities have been integrated: constraint setup, warm start, velocity iterations, position
// integration, position iterations, sleep.  `mc`: the manifolds the contact rows run over, [body][row] in row order --
// the world's own slots, or a packed copy of them (world_step below).
// The touching manifolds of a body, packed to the front in slot order (world_step_discrete, D::PACK_MANIFOLDS)
template <class D>
B2_FN void pack_manifolds(const World<D> &w, Manifold (&mc)[D::NB - D::FIRST_SOLVED][D::NSLOT])
{
    constexpr int NBS = D::NB - D::FIRST_SOLVED;
    B2_UNROLL
    for (int b = 0; b < NBS; ++b) {
        B2_UNROLL
        for (int r = 0; r < D::NSLOT; ++r) mc[b][r] = Manifold{-1, 0, 0, 0.0f, 0.0f, 0.0f, 0.0f, {0.0f, 0.0f}, {0.0f, 0.0f}, {0u, 0u}, {0.0f, 0.0f}, {0.0f, 0.0f}};
        int n = 0;
        B2_UNROLL
        for (int s = 0; s < D::NSLOT; ++s) {
            const bool touching = w.mf[b][s].count > 0;
            B2_UNROLL
            for (int r = 0; r <= s; ++r) {
                if (touching && n == r) mc[b][r] = w.mf[b][s];
            }
            n += touching ? 1 : 0;
        }
    }
}

// The same packing again for what the position iterations read of a manifold (type, count, local normal / point, the
// points), from the world where it lies in memory.  Between the velocity iterations' set-up and the position
// iterations these ten values per row are not used; read a second time behind a compiler barrier they do not occupy
// registers during the 180 velocity iterations (16 rows x 10 registers of the walker's 512).  Same values, same results.
template <class D>
B2_FN void repack_geometry(const World<D> &w, Manifold (&mc)[D::NB - D::FIRST_SOLVED][D::NSLOT])
{
    constexpr int NBS = D::NB - D::FIRST_SOLVED;
    asm volatile("" ::: "memory");
    B2_UNROLL
    for (int b = 0; b < NBS; ++b) {
        B2_UNROLL
        for (int r = 0; r < D::NSLOT; ++r) {                  // (every field anew: nothing of the first copy stays alive)
            Manifold &c = mc[b][r];
            c.count = 0; c.type = 0; c.lnx = 0.0f; c.lny = 0.0f; c.lpx = 0.0f; c.lpy = 0.0f;
            c.px[0] = 0.0f; c.px[1] = 0.0f; c.py[0] = 0.0f; c.py[1] = 0.0f;
        }
        int n = 0;
        B2_UNROLL
        for (int s = 0; s < D::NSLOT; ++s) {
            const Manifold &m = w.mf[b][s];
            const bool touching = m.count > 0;
            B2_UNROLL
            for (int r = 0; r <= s; ++r) {
                if (touching && n == r) {
                    Manifold &c = mc[b][r];
                    c.count = m.count; c.type = m.type; c.lnx = m.lnx; c.lny = m.lny; c.lpx = m.lpx; c.lpy = m.lpy;
                    c.px[0] = m.px[0]; c.px[1] = m.px[1]; c.py[0] = m.py[0]; c.py[1] = m.py[1];
                }
            }
            n += touching ? 1 : 0;
        }
    }
}

// one velocity iteration of the island: the joints in order, then the contact rows body by body
template <class D, bool REPACK>
B2_FN void velocity_iteration(Body (&body)[D::NB], Joint (&joint)[D::NJ], const JointTmp (&jt)[D::NJ], const JointRare *rare,
                              Manifold (&mc)[D::NB - D::FIRST_SOLVED][D::NSLOT],
                              const ContactTmp (&ct)[D::NB - D::FIRST_SOLVED][D::NSLOT], bool any_contact)
{
    B2_UNROLL
    for (int j = 0; j < D::NJ; ++j) joint_solve_velocity<D>(body, joint, j, jt[j], rare);
    if (any_contact) {
        B2_UNROLL
        for (int b = D::FIRST_SOLVED; b < D::NB; ++b) {
            // packed rows: the first empty row of a body ends its rows (one test for a body in the air instead of NSLOT)
            bool more = true;
            B2_UNROLL
            for (int r = 0; r < D::NSLOT; ++r) {
                if (REPACK) more = more && ct[b - D::FIRST_SOLVED][r].vcount != 0;
                if (more) contact_solve_velocity(mc[b - D::FIRST_SOLVED][r], ct[b - D::FIRST_SOLVED][r], body[b], D::body()[b]);
            }
        }
    }
}

template <class D, bool REPACK = false>
B2_FN void world_solve(World<D> &w, Manifold (&mc)[D::NB - D::FIRST_SOLVED][D::NSLOT], float dt)
{
    constexpr int NBS = D::NB - D::FIRST_SOLVED;
    // The solver runs on copies of the bodies and joints that are indexed by constants only: the world itself is indexed
    // by lane-dependent values elsewhere (manifold slots, the body of a time-of-impact event), which keeps it in memory
    // on the device -- and every velocity update of the 180 iterations would be a store to it.
    Body body[D::NB];
    Joint joint[D::NJ];
    B2_UNROLL
    for (int b = 0; b < D::NB; ++b) body[b] = w.body[b];
    B2_UNROLL
    for (int j = 0; j < D::NJ; ++j) joint[j] = w.joint[j];
    ContactTmp ct[NBS][D::NSLOT];
    JointTmp jt[D::NJ];
    JointRare rare_mem[D::NJ];
    JointRare *rare = rare_mem;
    B2_OPAQUE_PTR(rare);
    bool any_contact = false;
    B2_UNROLL
    for (int b = D::FIRST_SOLVED; b < D::NB; ++b) {
        B2_UNROLL
        for (int r = 0; r < D::NSLOT; ++r) {
            contact_init(mc[b - D::FIRST_SOLVED][r], ct[b - D::FIRST_SOLVED][r], body[b], D::body()[b], w.xf[b]);
            any_contact = any_contact || mc[b - D::FIRST_SOLVED][r].count > 0;
        }
    }
    B2_UNROLL
    for (int b = D::FIRST_SOLVED; b < D::NB; ++b) {
        B2_UNROLL
        for (int r = 0; r < D::NSLOT; ++r) contact_warm_start(mc[b - D::FIRST_SOLVED][r], ct[b - D::FIRST_SOLVED][r], body[b], D::body()[b]);
    }
    B2_UNROLL
    for (int j = 0; j < D::NJ; ++j) joint_init<D>(body, joint, w.xf, j, jt[j], rare, dt);

    B2_PHASE(2);
#ifdef SES_PHASE_TIMERS
    {                                                          // development build: which contact rows this env's iterations execute
        unsigned int rows = 0u;
        B2_UNROLL
        for (int b = 0; b < NBS; ++b) {
            bool more = true;
            B2_UNROLL
            for (int r = 0; r < D::NSLOT; ++r) {
                if (REPACK) more = more && ct[b][r].vcount != 0;
                if (more && ct[b][r].vcount != 0) rows |= 1u << (b * D::NSLOT + r);
            }
        }
        B2_PHASE_ROWS(any_contact ? rows : 0u, D::VEL_ITERS);
    }
#endif
    // (no wave-level votes anywhere in this file: a world only ever looks at its own state, so the code may run
    //  under any divergence; in flight the contact rows are skipped as a whole)
    const auto iterate = [&]() __attribute__((always_inline)) {
        velocity_iteration<D, REPACK>(body, joint, jt, rare, mc, ct, any_contact);
    };
    if constexpr (D::VEL_FIXED_POINT_CHECK < 0) {
        // (the body of velocity_iteration written out: through the helper the walker's loop comes out of hipcc with a third
        //  more AGPR moves -- 392 instead of 296 -- for the same source)
        for (int it = 0; it < D::VEL_ITERS; ++it) {
            B2_UNROLL
            for (int j = 0; j < D::NJ; ++j) joint_solve_velocity<D>(body, joint, j, jt[j], rare);
#ifdef SES_PHASE_SPLIT_VEL
            B2_PHASE(3);                                       // (development build) the joints of this iteration
#endif
            if (any_contact) {
                B2_UNROLL
                for (int b = D::FIRST_SOLVED; b < D::NB; ++b) {
                    bool more = true;
                    B2_UNROLL
                    for (int r = 0; r < D::NSLOT; ++r) {
                        if (REPACK) more = more && ct[b - D::FIRST_SOLVED][r].vcount != 0;
                        if (more) contact_solve_velocity(mc[b - D::FIRST_SOLVED][r], ct[b - D::FIRST_SOLVED][r], body[b], D::body()[b]);
                    }
                }
            }
#ifdef SES_PHASE_SPLIT_VEL
            B2_PHASE(16);                                      // its contact rows
#endif
        }
    } else {
        // Three plain loops instead of one with a test inside: the iterations before the check, the checked one, the rest.
        for (int it = 0; it < D::VEL_FIXED_POINT_CHECK; ++it) iterate();
        // what the checked iteration starts from ...
        Body body_was[D::NB];
        Joint joint_was[D::NJ];
        float imp_was[NBS][D::NSLOT][4];
        B2_UNROLL
        for (int b = 0; b < D::NB; ++b) body_was[b] = body[b];
        B2_UNROLL
        for (int j = 0; j < D::NJ; ++j) joint_was[j] = joint[j];
        B2_UNROLL
        for (int b = 0; b < NBS; ++b) {
            B2_UNROLL
            for (int r = 0; r < D::NSLOT; ++r) {
                imp_was[b][r][0] = mc[b][r].ni[0]; imp_was[b][r][1] = mc[b][r].ni[1];
                imp_was[b][r][2] = mc[b][r].ti[0]; imp_was[b][r][3] = mc[b][r].ti[1];
            }
        }
        iterate();
        // ... and what it ends with: the same bits (-0 is not +0 here) = a fixed point of the map = the result of all
        // D::VEL_ITERS iterations.  (A lane that skips the rest waits for its wave-mates; no votes.)
        uint32_t diff = 0u;
        B2_UNROLL
        for (int b = 0; b < D::NB; ++b)
            diff |= (B2_F2U(body_was[b].vx) ^ B2_F2U(body[b].vx)) | (B2_F2U(body_was[b].vy) ^ B2_F2U(body[b].vy)) |
                    (B2_F2U(body_was[b].w) ^ B2_F2U(body[b].w));
        B2_UNROLL
        for (int j = 0; j < D::NJ; ++j)
            diff |= (B2_F2U(joint_was[j].ix) ^ B2_F2U(joint[j].ix)) | (B2_F2U(joint_was[j].iy) ^ B2_F2U(joint[j].iy)) |
                    (B2_F2U(joint_was[j].iz) ^ B2_F2U(joint[j].iz)) | (B2_F2U(joint_was[j].im) ^ B2_F2U(joint[j].im));
        if (any_contact) {
            B2_UNROLL
            for (int b = 0; b < NBS; ++b) {
                B2_UNROLL
                for (int r = 0; r < D::NSLOT; ++r)
                    diff |= (B2_F2U(imp_was[b][r][0]) ^ B2_F2U(mc[b][r].ni[0])) | (B2_F2U(imp_was[b][r][1]) ^ B2_F2U(mc[b][r].ni[1])) |
                            (B2_F2U(imp_was[b][r][2]) ^ B2_F2U(mc[b][r].ti[0])) | (B2_F2U(imp_was[b][r][3]) ^ B2_F2U(mc[b][r].ti[1]));
            }
        }
        if (diff != 0u) {
            for (int it = D::VEL_FIXED_POINT_CHECK + 1; it < D::VEL_ITERS; ++it) iterate();
        }
    }

    B2_PHASE(3);
    // integrate positions
    B2_UNROLL
    for (int b = 0; b < D::NB; ++b) {
        Body &B = body[b];
        const float trx = dt * B.vx, try_ = dt * B.vy;
        if (trx * trx + try_ * try_ > MAX_TRANSLATION_SQ) {
            const float ratio = MAX_TRANSLATION / B2_SQRT(trx * trx + try_ * try_);
            B.vx *= ratio; B.vy *= ratio;
        }
        const float rot = dt * B.w;
        if (rot * rot > MAX_ROTATION_SQ) {
            const float ratio = MAX_ROTATION / b2abs(rot);
            B.w *= ratio;
        }
        B.cx += dt * B.vx; B.cy += dt * B.vy;
        B.a += dt * B.w;
    }

    if constexpr (REPACK) repack_geometry(w, mc);
    B2_PHASE(4);

    // position iterations with Box2D's own early exit
    bool position_solved = false;
    for (int pit = 0; pit < D::POS_ITERS && !position_solved; ++pit) {
        {
            float min_separation = 0.0f;
            if (any_contact) {
                B2_UNROLL
                for (int b = D::FIRST_SOLVED; b < D::NB; ++b) {
                    bool more = true;
                    B2_UNROLL
                    for (int r = 0; r < D::NSLOT; ++r) {
                        if (REPACK) more = more && mc[b - D::FIRST_SOLVED][r].count != 0;
                        if (more)
                            min_separation = b2min(min_separation, contact_solve_position(mc[b - D::FIRST_SOLVED][r], body[b], D::body()[b]));
                    }
                }
            }
            const bool contacts_okay = min_separation >= -3.0f * LINEAR_SLOP;
            bool joints_okay = true;
            B2_UNROLL
            for (int j = 0; j < D::NJ; ++j) {
                const bool ok = joint_solve_position<D>(body, joint, j);
                joints_okay = joints_okay && ok;
            }
            position_solved = contacts_okay && joints_okay;
        }
    }

    B2_PHASE(5);
    // sleep (b2Island::Solve, the island = all bodies of the world)
    float min_sleep = FLT_BIG;
    B2_UNROLL
    for (int b = 0; b < D::NB; ++b) {
        const Body &B = body[b];
        if (B.w * B.w > ANGULAR_SLEEP_TOL * ANGULAR_SLEEP_TOL || B.vx * B.vx + B.vy * B.vy > LINEAR_SLEEP_TOL * LINEAR_SLEEP_TOL) {
            w.sleep_time[b] = 0.0f;
            min_sleep = 0.0f;
        } else {
            w.sleep_time[b] += dt;
            min_sleep = b2min(min_sleep, w.sleep_time[b]);
        }
    }
    if (min_sleep >= TIME_TO_SLEEP && position_solved) w.awake = false;
    B2_UNROLL
    for (int b = 0; b < D::NB; ++b) w.body[b] = body[b];
    B2_UNROLL
    for (int j = 0; j < D::NJ; ++j) w.joint[j] = joint[j];
}

// ------------------------------------------------------------------------------------------------------------------
// b2World::SolveTOI for worlds whose only static geometry is the terrain and whose dynamic bodies do not collide with
// each other: every TOI event concerns ONE body and the terrain edges near it.  After the discrete solve each body has
// a sweep (start of step -> solved end of step); while some (body, candidate edge) pair reaches the target separation
// before the end of the step, the earliest such body is rolled back to that time, its manifolds are updated
// (b2Contact::Update: begin / end events fire here too -- a hull that touches ends the episode), and b2Island::SolveTOI
// runs on it alone: up to 20 position iterations against its touching manifolds with b2_toiBaugarte, the sweep
// restarts there ("leap of faith"), the contacts' velocity rows are solved from zero impulses for all velocity
// iterations (joints are ignored in a sub-step, as in Box2D: their error is repaired by the next step), and the body
// moves on for the rest of the step with the velocity it is left with.  Impulses of a sub-step are not kept.
constexpr float TOI_BAUMGARTE = 0.75f;
constexpr int TOI_MAX_SUBSTEPS = 8;                  // b2_maxSubSteps, per contact and step

template <class D, class T>
B2_FN void world_solve_toi(World<D> &w, const T &terr, Sweep (&sw)[D::NB], float dt)
{
    int toi_count[D::NB][D::NSLOT];
    bool disabled[D::NB][D::NSLOT];
    B2_UNROLL
    for (int b = 0; b < D::NB; ++b) {
        B2_UNROLL
        for (int s = 0; s < D::NSLOT; ++s) { toi_count[b][s] = 0; disabled[b][s] = false; }
    }
    // Box2D caches a contact's time of impact until one of its bodies moves (b2Contact::m_toi, e_toiFlag): so does this
    // loop -- after an event only the body that took the sub-step has a new sweep, the others' pairs keep their alpha
    float alpha_of[D::NB][D::NSLOT];
    bool stale[D::NB];
    B2_UNROLL
    for (int b = 0; b < D::NB; ++b) stale[b] = true;
    for (int ev = 0; ev < D::NB * D::NSLOT * TOI_MAX_SUBSTEPS; ++ev) {
        float min_alpha = 1.0f;
        int min_b = -1, min_s = -1;
        B2_UNROLL
        for (int b = 0; b < D::NB; ++b) {
          if (stale[b]) {
            stale[b] = false;
            B2_UNROLL
            for (int s = 0; s < D::NSLOT; ++s) alpha_of[b][s] = 1.0f;
            const Poly &P = D::poly()[b];
            const BodyDef &bd = D::body()[b];
            // fattened AABB of the swept polygon (b2Fixture::Synchronize: the proxy covers both ends of the sweep);
            // (sx, sy): the polygon's vertices at the start of the sweep, r_max: the farthest vertex from the centre of mass
            float xmin = FLT_BIG, xmax = -FLT_BIG, ymin = FLT_BIG;
            float sx[6], sy[6], sx0 = FLT_BIG, sx1 = -FLT_BIG, sy0 = FLT_BIG, sy1 = -FLT_BIG, r_max_sq = 0.0f;
            B2_UNROLL
            for (int e = 0; e < 2; ++e) {
                Xf xf;
                sweep_xf(sw[b], bd, e ? 1.0f : 0.0f, xf);
                B2_UNROLL
                for (int i = 0; i < 6; ++i) {
                    if (i < P.n) {
                        const float x = (xf.c * P.vx[i] - xf.s * P.vy[i]) + xf.px, y = (xf.s * P.vx[i] + xf.c * P.vy[i]) + xf.py;
                        xmin = b2min(xmin, x); xmax = b2max(xmax, x); ymin = b2min(ymin, y);
                        if (e == 0) {
                            sx[i] = x; sy[i] = y;
                            sx0 = b2min(sx0, x); sx1 = b2max(sx1, x); sy0 = b2min(sy0, y); sy1 = b2max(sy1, y);
                            const float rx = P.vx[i] - bd.lcx, ry = P.vy[i] - bd.lcy;
                            r_max_sq = b2max(r_max_sq, rx * rx + ry * ry);
                        }
                    } else if (e == 0) {
                        sx[i] = 0.0f; sy[i] = 0.0f;
                    }
                }
            }
            // A pair that cannot come out of b2TimeOfImpact as "touching" is not evaluated (every other outcome leaves
            // alpha at 1).  No point of the polygon moves farther during the sweep than the centre's path plus r_max
            // times the turn (`reach`); at the start of the sweep the shapes are at least as far apart as (i) their
            // boxes and (ii) the polygon is from the edge's line, measured along the line's normal.  If that lower
            // bound exceeds reach + target + tolerance, the separation stays above target + tolerance for the whole
            // sweep.  This is what keeps a lander in flight and, above all, one RESTING on its legs (core shapes
            // 0.015 apart, not moving) from paying a GJK evaluation per leg, edge and step.
            const float dcx = sw[b].cx - sw[b].c0x, dcy = sw[b].cy - sw[b].c0y;
            const float reach = (B2_SQRT(dcx * dcx + dcy * dcy) + B2_SQRT(r_max_sq) * b2abs(sw[b].a - sw[b].a0)) +
                                ((b2max(LINEAR_SLOP, POLY_RADIUS + POLY_RADIUS - 3.0f * LINEAR_SLOP) + 0.25f * LINEAR_SLOP) + 2.0e-4f);
            int k_lo = terr.index_of(xmin - AABB_EXTENSION), k_hi = terr.index_of(xmax + AABB_EXTENSION);
            k_lo = k_lo < 0 ? 0 : k_lo;
            k_hi = k_hi > terr.n_edges() - 1 ? terr.n_edges() - 1 : k_hi;
            k_hi = k_hi > k_lo + D::NSLOT - 1 ? k_lo + D::NSLOT - 1 : k_hi;
            B2_UNROLL
            for (int s = 0; s < D::NSLOT; ++s) {
                const int k = k_lo + ((s - k_lo) & (D::NSLOT - 1));
                if (k <= k_hi && k_lo <= k_hi && !disabled[b][s] && toi_count[b][s] <= TOI_MAX_SUBSTEPS) {
                    ToiPair pr;
                    terr.edge(k, pr.ex[0], pr.ey[0], pr.ex[1], pr.ey[1]);
                    pr.P = &P;
                    const float gx = b2max(0.0f, b2max(sx0 - b2max(pr.ex[0], pr.ex[1]), b2min(pr.ex[0], pr.ex[1]) - sx1));
                    const float gy = b2max(0.0f, b2max(sy0 - b2max(pr.ey[0], pr.ey[1]), b2min(pr.ey[0], pr.ey[1]) - sy1));
                    float line_sep = 0.0f;
                    {
                        float nx = pr.ey[1] - pr.ey[0], ny = -(pr.ex[1] - pr.ex[0]);     // the edge's normal, either sense
                        const float len = B2_SQRT(nx * nx + ny * ny);
                        float lo = FLT_BIG, hi = -FLT_BIG;
                        B2_UNROLL
                        for (int i = 0; i < 6; ++i) {
                            if (i < P.n) {
                                const float d = nx * (sx[i] - pr.ex[0]) + ny * (sy[i] - pr.ey[0]);
                                lo = b2min(lo, d); hi = b2max(hi, d);
                            }
                        }
                        // all vertices on one side of the line: its distance (un-normalised d / len, rounded down a little)
                        if (len > 1.0e-6f) line_sep = b2max(0.0f, b2max(lo, -hi)) / len * 0.999f;
                    }
                    const bool out_of_reach = gx * gx + gy * gy > reach * reach || line_sep > reach;
                    if (!(ymin - AABB_EXTENSION > b2max(pr.ey[0], pr.ey[1])) && !out_of_reach) {
                        float t;
                        const int state = time_of_impact(pr, sw[b], bd, t);
                        alpha_of[b][s] = state == TOI_TOUCHING ? b2min(sw[b].alpha0 + (1.0f - sw[b].alpha0) * t, 1.0f) : 1.0f;
                    }
                }
            }
          }
            B2_UNROLL
            for (int s = 0; s < D::NSLOT; ++s) {
                if (alpha_of[b][s] < min_alpha) { min_alpha = alpha_of[b][s]; min_b = b; min_s = s; }
            }
        }
        B2_PHASE(14);
        if (min_b < 0 || 1.0f - 10.0f * 1.1920928955078125e-7f < min_alpha) break;
        bool stop = false;
        B2_UNROLL
        for (int b = 0; b < D::NB; ++b) {
            if (b == min_b) {
                stale[b] = true;
                const BodyDef &bd = D::body()[b];
                const Sweep backup = sw[b];
                sweep_advance(sw[b], min_alpha);
                w.body[b].cx = sw[b].c0x; w.body[b].cy = sw[b].c0y; w.body[b].a = sw[b].a0;
                collide_body(w, terr, b);                                    // b2Contact::Update at the time of impact
                bool touching = false;
                B2_UNROLL
                for (int s = 0; s < D::NSLOT; ++s) {
                    if (s == min_s) {
                        toi_count[b][s] += 1;
                        if (b >= D::FIRST_SOLVED) touching = w.mf[b >= D::FIRST_SOLVED ? b - D::FIRST_SOLVED : 0][s].count > 0;
                    }
                }
                if (b < D::FIRST_SOLVED) {
                    // the hull: a touch has just ended the episode (collide_body set game_over); whatever a sub-step
                    // would do to it is never observed
                    touching = w.game_over;
                    stop = touching;
                }
                if (!touching) {                                             // "the contact is not touching at the TOI": disabled for this step
                    B2_UNROLL
                    for (int s = 0; s < D::NSLOT; ++s) {
                        if (s == min_s) disabled[b][s] = true;
                    }
                    sw[b] = backup;
                    w.body[b].cx = sw[b].cx; w.body[b].cy = sw[b].cy; w.body[b].a = sw[b].a;
                } else if (b >= D::FIRST_SOLVED) {
                    // b2Island::SolveTOI on this body alone
                    const int bi = b >= D::FIRST_SOLVED ? b - D::FIRST_SOLVED : 0;
                    // the sub-step works on copies of the body and of its manifolds (constant indices: registers on the
                    // device, not the world's memory -- its loops would store the body after every row)
                    Body B = w.body[b];
                    Manifold mt[D::NSLOT];
                    B2_UNROLL
                    for (int s = 0; s < D::NSLOT; ++s) {
                        mt[s] = w.mf[bi][s];
                        mt[s].ni[0] = 0.0f; mt[s].ni[1] = 0.0f; mt[s].ti[0] = 0.0f; mt[s].ti[1] = 0.0f;   // no warm starting
                    }
                    for (int i = 0; i < 20; ++i) {
                        float min_separation = 0.0f;
                        B2_UNROLL
                        for (int s = 0; s < D::NSLOT; ++s)
                            min_separation = b2min(min_separation, contact_solve_position(mt[s], B, bd, TOI_BAUMGARTE));
                        if (min_separation >= -1.5f * LINEAR_SLOP) break;
                    }
                    sw[b].c0x = B.cx; sw[b].c0y = B.cy; sw[b].a0 = B.a;     // leap of faith to the new safe state
                    Xf xf;
                    xf_of(B, bd, xf);
                    ContactTmp ct[D::NSLOT];
                    B2_UNROLL
                    for (int s = 0; s < D::NSLOT; ++s) contact_init(mt[s], ct[s], B, bd, xf);
                    // all D::VEL_ITERS iterations, as in Box2D (subStep.velocityIterations = step.velocityIterations) -- but
                    // an iteration that leaves the body's velocity and every accumulated impulse bit for bit where they
                    // were is a fixed point of the map, and so are all the iterations after it: stop there.  (One body
                    // against one or two manifolds, no joints, no warm start: that happens after ~25 iterations.)
                    for (int it = 0; it < D::VEL_ITERS; ++it) {
                        const float vx0 = B.vx, vy0 = B.vy, w0 = B.w;
                        float imp0[D::NSLOT][4];
                        B2_UNROLL
                        for (int s = 0; s < D::NSLOT; ++s) {
                            imp0[s][0] = mt[s].ni[0]; imp0[s][1] = mt[s].ni[1]; imp0[s][2] = mt[s].ti[0]; imp0[s][3] = mt[s].ti[1];
                        }
                        B2_UNROLL
                        for (int s = 0; s < D::NSLOT; ++s) contact_solve_velocity(mt[s], ct[s], B, bd);
                        uint32_t diff = (B2_F2U(vx0) ^ B2_F2U(B.vx)) | (B2_F2U(vy0) ^ B2_F2U(B.vy)) | (B2_F2U(w0) ^ B2_F2U(B.w));
                        B2_UNROLL
                        for (int s = 0; s < D::NSLOT; ++s)
                            diff |= (B2_F2U(imp0[s][0]) ^ B2_F2U(mt[s].ni[0])) | (B2_F2U(imp0[s][1]) ^ B2_F2U(mt[s].ni[1])) |
                                    (B2_F2U(imp0[s][2]) ^ B2_F2U(mt[s].ti[0])) | (B2_F2U(imp0[s][3]) ^ B2_F2U(mt[s].ti[1]));
#ifndef B2_RUN_ALL_ITERATIONS
                        if (diff == 0u) break;                              // bitwise: -0 is not +0, as in the main loop
#else
                        (void)diff;
#endif
                    }
                    const float h = (1.0f - min_alpha) * dt;
                    const float trx = h * B.vx, try_ = h * B.vy;
                    if (trx * trx + try_ * try_ > MAX_TRANSLATION_SQ) {
                        const float ratio = MAX_TRANSLATION / B2_SQRT(trx * trx + try_ * try_);
                        B.vx *= ratio; B.vy *= ratio;
                    }
                    const float rot = h * B.w;
                    if (rot * rot > MAX_ROTATION_SQ) {
                        const float ratio = MAX_ROTATION / b2abs(rot);
                        B.w *= ratio;
                    }
                    B.cx += h * B.vx; B.cy += h * B.vy;
                    B.a += h * B.w;
                    sw[b].cx = B.cx; sw[b].cy = B.cy; sw[b].a = B.a;
                    w.body[b] = B;
                }
            }
        }
        B2_PHASE(15);
        if (stop) break;
    }
}

template <class D, class T>
B2_FN void world_step_discrete(World<D> &w, const T &terr, float dt, Sweep (&sw)[D::NB])
{
    if constexpr (D::CONTINUOUS) {
        B2_UNROLL
        for (int b = 0; b < D::NB; ++b) {
            sw[b].c0x = w.body[b].cx; sw[b].c0y = w.body[b].cy; sw[b].a0 = w.body[b].a; sw[b].alpha0 = 0.0f;
        }
    }
    B2_PHASE(0);
    collide(w, terr);
    B2_PHASE(1);

    // b2Island::Solve: integrate velocities (gravity, no damping: v *= 1 / (1 + h * 0) is exact)
    B2_UNROLL
    for (int b = 0; b < D::NB; ++b) {
        const BodyDef &bd = D::body()[b];
        const float fx = b == 0 ? w.fx : 0.0f, fy = b == 0 ? w.fy : 0.0f;
        w.body[b].vx += dt * (0.0f + bd.inv_mass * fx);
        w.body[b].vy += dt * (D::GRAVITY_Y + bd.inv_mass * fy);
    }
    w.fx = 0.0f; w.fy = 0.0f;

    if constexpr (!D::PACK_MANIFOLDS) {
        world_solve(w, w.mf, dt);
    } else {
        // The touching manifolds of a body, packed to the front in slot order: the solver's row order is unchanged (it
        // skipped the empty slots anyway), so are the results.  Which slot a manifold sits in depends on where along
        // the terrain the body stands (edge mod NSLOT); when the lanes of a wavefront carry different worlds,
        // slot-indexed rows make the wave execute the union of everybody's slots -- packed, it executes
        // max-over-lanes(count) rows per body.  Worth the copies for the walker (4 x 4 slots, a foot touches one or
        // two edges: 2.5x on the rollout), not for the lander (2 x 2 slots).
        constexpr int NBS = D::NB - D::FIRST_SOLVED;
        Manifold mc[NBS][D::NSLOT];
        pack_manifolds(w, mc);
        world_solve<D, true>(w, mc, dt);
        // the accumulated impulses go back to their slots (next step's warm start, b2Contact::Update in collide())
        B2_UNROLL
        for (int b = 0; b < NBS; ++b) {
            int n = 0;
            B2_UNROLL
            for (int s = 0; s < D::NSLOT; ++s) {
                const bool touching = w.mf[b][s].count > 0;
                B2_UNROLL
                for (int r = 0; r <= s; ++r) {
                    if (touching && n == r) {
                        w.mf[b][s].ni[0] = mc[b][r].ni[0]; w.mf[b][s].ni[1] = mc[b][r].ni[1];
                        w.mf[b][s].ti[0] = mc[b][r].ti[0]; w.mf[b][s].ti[1] = mc[b][r].ti[1];
                    }
                }
                n += touching ? 1 : 0;
            }
        }
    }
    if constexpr (D::CONTINUOUS) {
        B2_UNROLL
        for (int b = 0; b < D::NB; ++b) { sw[b].cx = w.body[b].cx; sw[b].cy = w.body[b].cy; sw[b].a = w.body[b].a; }
    }
    B2_PHASE(6);
}

// The two halves of b2World::Step are separate functions because the device build keeps them in separate REAL
// functions: the discrete half holds the whole world in registers and contains no call; the continuous half works on
// the world where it lies in memory, reads little more than the sweeps in a step without an event, and is the only one
// that calls time_of_impact.  (As one function, the values that had to survive the call sites -- the whole world --
// were parked in scratch memory in every step: +30 % on the lander rollout.)
template <class D, class T>
B2_FN void world_step_toi(World<D> &w, const T &terr, float dt, Sweep (&sw)[D::NB])
{
    B2_PHASE(7);
    if constexpr (D::CONTINUOUS) {
        if (w.awake) world_solve_toi(w, terr, sw, dt);
    }
    B2_PHASE(8);
}

template <class D, class T>
B2_FN void world_step(World<D> &w, const T &terr, float dt)
{
    Sweep sw[D::NB];
    world_step_discrete(w, terr, dt, sw);
    world_step_toi(w, terr, dt, sw);
}

}  // namespace b2l

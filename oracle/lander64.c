/* lander64.c -- TEST INFRASTRUCTURE (oracle).  An INDEPENDENTLY WRITTEN float64 integration of gym's LunarLanderContinuous-v2,
 * used only by tests/test_oracle_lander.py to bound what the product's float32 Box2D-style world (simple-es_amd/csrc/ses_b2.h,
 * ses_lander_env.h) may get wrong: it shares no code and no formulas with that world.
 *
 * What is the same by construction: the INPUTS (the 16 reset uniforms, the per-step engine dispersion numbers, the action
 * sequence), gym's env rules (lunar_lander.py: engine impulses, observation, shaping reward, termination) and Box2D's
 * documented tolerances (linear / angular slop, sleep thresholds).  What is different on purpose:
 *   - double precision, libm sin / cos;
 *   - mass, centroid and inertia of every body computed here from the polygons (not read from ses_b2_shapes.h);
 *   - ONE generic constraint row type (two bodies, six Jacobian coefficients, bounds) for joint points, joint limits, joint
 *     motors, contact normals and friction, instead of b2RevoluteJoint's 3x3 block and b2ContactSolver's two-point block;
 *   - the velocity constraints are solved to CONVERGENCE (projected Gauss-Seidel until no impulse moves by more than
 *     1e-13, up to 20 000 sweeps) where Box2D stops after 180 iterations;
 *   - contacts are vertex-versus-terrain-line tests (no clipping, no manifold ids, no warm-starting across steps, no
 *     time-of-impact pass: a fast body may sink in for a step and is pushed out by the position correction);
 *   - position errors are removed by repeated projection until they are inside Box2D's slops.
 * The envelope the float32 trajectories are held to against this integration is stated in the test.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define NBODY 3
#define MAXROWS 40

static const double SCALE = 30.0, FPS = 50.0, VW = 600.0 / 30.0, VH = 400.0 / 30.0;
static const double MAIN_POWER = 13.0, SIDE_POWER = 0.6, LEG_AWAY = 20.0, LEG_DOWN = 18.0, LEG_W = 2.0, LEG_H = 8.0;
static const double LEG_TORQUE = 40.0, SIDE_H = 14.0, SIDE_AWAY = 12.0;
static const double LIN_SLOP = 0.005, ANG_SLOP = 2.0 / 180.0 * 3.14159265358979323846, SKIN = 0.02; /* 2 x b2_polygonRadius */
static const double SLEEP_LIN = 0.01, SLEEP_ANG = 2.0 / 180.0 * 3.14159265358979323846, SLEEP_TIME = 0.5;

typedef struct {
    double px, py, a;          /* body origin and angle */
    double vx, vy, w;          /* velocity of the centre of mass, angular velocity */
    double im, ii;             /* inverse mass, inverse inertia about the centre of mass */
    double lcx, lcy;           /* centre of mass in body coordinates */
    int nv;
    double vx_[6], vy_[6];     /* polygon, body coordinates */
    double mu;
} Body;

typedef struct {
    Body b[NBODY];
    double ty[11];
    double fx, fy;             /* force on the hull, acts during the next world step */
    double sleep_time;
    int awake, game_over, leg_contact[2];
    double prev_shaping;
    int has_prev;
    double helipad_y;
} L64;

typedef struct {
    int a, b;                  /* body indices; a = -1: the world */
    double ja[3], jb[3];       /* Jacobian: Cdot = ja . (va, wa) + jb . (vb, wb) - target */
    double target, lo, hi, lam, k;
    int friction_of;           /* >= 0: bounds are +-mu * lam of that row */
    double mu;
} Row;

static void com(const Body *B, double *cx, double *cy)
{
    const double s = sin(B->a), c = cos(B->a);
    *cx = B->px + c * B->lcx - s * B->lcy;
    *cy = B->py + s * B->lcx + c * B->lcy;
}

static void set_polygon(Body *B, int n, const double *x, const double *y, double density, double mu)
{
    /* area, centroid and second moment by the shoelace formulas */
    double A = 0, cx = 0, cy = 0, J = 0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1) % n;
        const double cr = x[i] * y[j] - x[j] * y[i];
        A += cr;
        cx += (x[i] + x[j]) * cr;
        cy += (y[i] + y[j]) * cr;
        J += cr * (x[i] * x[i] + x[i] * x[j] + x[j] * x[j] + y[i] * y[i] + y[i] * y[j] + y[j] * y[j]);
    }
    A *= 0.5;
    cx /= 6.0 * A;
    cy /= 6.0 * A;
    const double mass = density * A;
    const double I0 = density * J / 12.0;                    /* about the body origin */
    const double Ic = I0 - mass * (cx * cx + cy * cy);
    B->im = 1.0 / mass;
    B->ii = 1.0 / Ic;
    B->lcx = cx; B->lcy = cy;
    B->nv = n;
    for (int i = 0; i < n; ++i) { B->vx_[i] = x[i]; B->vy_[i] = y[i]; }
    B->mu = mu;
}

/* signed distance of a point from the terrain line under it (positive = above), and that segment's unit normal */
static double terrain_distance(const L64 *s, double x, double y, double *nx, double *ny)
{
    int k = (int)floor(x * 0.5);
    if (k < 0) k = 0;
    if (k > 9) k = 9;
    const double x1 = 2.0 * k, y1 = s->ty[k], x2 = 2.0 * (k + 1), y2 = s->ty[k + 1];
    const double ex = x2 - x1, ey = y2 - y1, len = sqrt(ex * ex + ey * ey);
    *nx = -ey / len; *ny = ex / len;
    return (x - x1) * *nx + (y - y1) * *ny;
}

static void apply(Body *B, const double *j, double dl)
{
    B->vx += B->im * j[0] * dl; B->vy += B->im * j[1] * dl; B->w += B->ii * j[2] * dl;
}

static double row_k(const L64 *s, const Row *r)
{
    double k = 0;
    if (r->a >= 0) { const Body *A = &s->b[r->a]; k += A->im * (r->ja[0] * r->ja[0] + r->ja[1] * r->ja[1]) + A->ii * r->ja[2] * r->ja[2]; }
    { const Body *B = &s->b[r->b]; k += B->im * (r->jb[0] * r->jb[0] + r->jb[1] * r->jb[1]) + B->ii * r->jb[2] * r->jb[2]; }
    return k;
}

/* row for "velocity of point P of body b minus velocity of point P of body a, along direction d" */
static void point_row(const L64 *s, Row *r, int a, int b, double px, double py, double dx, double dy)
{
    memset(r, 0, sizeof *r);
    r->a = a; r->b = b; r->friction_of = -1;
    double cx, cy;
    com(&s->b[b], &cx, &cy);
    r->jb[0] = dx; r->jb[1] = dy; r->jb[2] = (px - cx) * dy - (py - cy) * dx;
    if (a >= 0) {
        com(&s->b[a], &cx, &cy);
        r->ja[0] = -dx; r->ja[1] = -dy; r->ja[2] = -((px - cx) * dy - (py - cy) * dx);
    }
    r->lo = -1e300; r->hi = 1e300;
}

static void joint_anchor(const L64 *s, int leg, double *ax, double *ay)   /* the hull's origin: localAnchorA = (0, 0) */
{
    (void)leg;
    *ax = s->b[0].px; *ay = s->b[0].py;
}

static void leg_anchor(const L64 *s, int leg, double *ax, double *ay)     /* localAnchorB = (i * LEG_AWAY, LEG_DOWN) / SCALE */
{
    const Body *B = &s->b[1 + leg];
    const double i = leg == 0 ? -1.0 : 1.0, lx = i * LEG_AWAY / SCALE, ly = LEG_DOWN / SCALE;
    const double sn = sin(B->a), c = cos(B->a);
    *ax = B->px + c * lx - sn * ly; *ay = B->py + sn * lx + c * ly;
}

static void limits(int leg, double *lo, double *hi)
{
    if (leg == 0) { *lo = 0.9 - 0.5; *hi = 0.9; } else { *lo = -0.9; *hi = -0.9 + 0.5; }
}

static int build_rows(L64 *s, Row *rows, double dt, int *touch_leg, int *touch_hull)
{
    int n = 0;
    for (int leg = 0; leg < 2; ++leg) {
        double ax, ay;
        leg_anchor(s, leg, &ax, &ay);
        point_row(s, &rows[n++], 0, 1 + leg, ax, ay, 1.0, 0.0);
        point_row(s, &rows[n++], 0, 1 + leg, ax, ay, 0.0, 1.0);
        /* motor: relative angular velocity -> motorSpeed = 0.3 * i, |impulse| <= dt * maxMotorTorque */
        Row *m = &rows[n++];
        memset(m, 0, sizeof *m);
        m->a = 0; m->b = 1 + leg; m->friction_of = -1;
        m->ja[2] = -1.0; m->jb[2] = 1.0;
        m->target = 0.3 * (leg == 0 ? -1.0 : 1.0);
        m->lo = -dt * LEG_TORQUE; m->hi = dt * LEG_TORQUE;
        /* limit: active at or beyond a bound */
        double lo, hi;
        limits(leg, &lo, &hi);
        const double ang = s->b[1 + leg].a - s->b[0].a;
        if (ang <= lo || ang >= hi) {
            Row *l = &rows[n++];
            memset(l, 0, sizeof *l);
            l->a = 0; l->b = 1 + leg; l->friction_of = -1;
            l->ja[2] = -1.0; l->jb[2] = 1.0;
            if (ang <= lo) { l->lo = 0.0; l->hi = 1e300; } else { l->lo = -1e300; l->hi = 0.0; }
        }
    }
    *touch_hull = 0;
    touch_leg[0] = touch_leg[1] = 0;
    for (int b = 0; b < NBODY; ++b) {
        const Body *B = &s->b[b];
        const double sn = sin(B->a), c = cos(B->a);
        for (int i = 0; i < B->nv; ++i) {
            const double x = B->px + c * B->vx_[i] - sn * B->vy_[i], y = B->py + sn * B->vx_[i] + c * B->vy_[i];
            double nx, ny;
            const double d = terrain_distance(s, x, y, &nx, &ny);
            if (d < SKIN) {
                if (b == 0) { *touch_hull = 1; continue; }           /* the hull touching ends the episode: no row needed */
                touch_leg[b - 1] = 1;
                if (n + 2 > MAXROWS) continue;
                Row *nr = &rows[n];
                point_row(s, nr, -1, b, x, y, nx, ny);
                nr->lo = 0.0; nr->hi = 1e300;
                Row *fr = &rows[n + 1];
                point_row(s, fr, -1, b, x, y, ny, -nx);
                fr->friction_of = n; fr->mu = sqrt(B->mu * 0.1);      /* b2MixFriction with the terrain's 0.1 */
                n += 2;
            }
        }
    }
    for (int i = 0; i < n; ++i) rows[i].k = row_k(s, &rows[i]);
    return n;
}

static void solve_velocity(L64 *s, Row *rows, int n)
{
    for (int sweep = 0; sweep < 20000; ++sweep) {
        double moved = 0.0;
        for (int i = 0; i < n; ++i) {
            Row *r = &rows[i];
            double cdot = -r->target;
            if (r->a >= 0) { const Body *A = &s->b[r->a]; cdot += r->ja[0] * A->vx + r->ja[1] * A->vy + r->ja[2] * A->w; }
            { const Body *B = &s->b[r->b]; cdot += r->jb[0] * B->vx + r->jb[1] * B->vy + r->jb[2] * B->w; }
            double lo = r->lo, hi = r->hi;
            if (r->friction_of >= 0) { hi = r->mu * rows[r->friction_of].lam; lo = -hi; }
            double lam = r->lam - cdot / r->k;
            if (lam < lo) lam = lo;
            if (lam > hi) lam = hi;
            const double dl = lam - r->lam;
            r->lam = lam;
            if (r->a >= 0) apply(&s->b[r->a], r->ja, dl);
            apply(&s->b[r->b], r->jb, dl);
            if (fabs(dl) > moved) moved = fabs(dl);
        }
        if (moved < 1e-13) break;
    }
}

/* remove position errors by repeated projection: joint anchors, joint limits, penetrations (inside Box2D's slops) */
static void solve_position(L64 *s)
{
    for (int it = 0; it < 200; ++it) {
        double worst = 0.0;
        for (int leg = 0; leg < 2; ++leg) {
            Body *A = &s->b[0], *B = &s->b[1 + leg];
            double lo, hi;
            limits(leg, &lo, &hi);
            const double ang = B->a - A->a;
            double C = 0.0;
            if (ang < lo - ANG_SLOP) C = ang - (lo - ANG_SLOP); else if (ang > hi + ANG_SLOP) C = ang - (hi + ANG_SLOP);
            if (C != 0.0) {
                const double lam = -C / (A->ii + B->ii);
                /* rotate about the centres of mass: the origins move with them */
                double cx, cy;
                com(A, &cx, &cy); A->a -= A->ii * lam; { const double sn = sin(A->a), c = cos(A->a); A->px = cx - (c * A->lcx - sn * A->lcy); A->py = cy - (sn * A->lcx + c * A->lcy); }
                com(B, &cx, &cy); B->a += B->ii * lam; { const double sn = sin(B->a), c = cos(B->a); B->px = cx - (c * B->lcx - sn * B->lcy); B->py = cy - (sn * B->lcx + c * B->lcy); }
                if (fabs(C) > worst) worst = fabs(C) * 0.1;
            }
            double ax, ay, bx, by, cax, cay, cbx, cby;
            joint_anchor(s, leg, &ax, &ay);
            leg_anchor(s, leg, &bx, &by);
            const double ex = bx - ax, ey = by - ay, err = sqrt(ex * ex + ey * ey);
            if (err > 1e-12) {
                com(A, &cax, &cay); com(B, &cbx, &cby);
                const double dx = ex / err, dy = ey / err;
                const double ra = (ax - cax) * dy - (ay - cay) * dx, rb = (bx - cbx) * dy - (by - cby) * dx;
                const double k = A->im + B->im + A->ii * ra * ra + B->ii * rb * rb, lam = -err / k;
                cax -= A->im * dx * lam; cay -= A->im * dy * lam; A->a -= A->ii * ra * lam;
                cbx += B->im * dx * lam; cby += B->im * dy * lam; B->a += B->ii * rb * lam;
                { const double sn = sin(A->a), c = cos(A->a); A->px = cax - (c * A->lcx - sn * A->lcy); A->py = cay - (sn * A->lcx + c * A->lcy); }
                { const double sn = sin(B->a), c = cos(B->a); B->px = cbx - (c * B->lcx - sn * B->lcy); B->py = cby - (sn * B->lcx + c * B->lcy); }
                if (err > worst) worst = err;
            }
        }
        for (int b = 1; b < NBODY; ++b) {
            Body *B = &s->b[b];
            for (int i = 0; i < B->nv; ++i) {
                const double sn = sin(B->a), c = cos(B->a);
                const double x = B->px + c * B->vx_[i] - sn * B->vy_[i], y = B->py + sn * B->vx_[i] + c * B->vy_[i];
                double nx, ny;
                const double sep = terrain_distance(s, x, y, &nx, &ny) - SKIN;
                if (sep < -LIN_SLOP) {
                    double C = 0.2 * (sep + LIN_SLOP);                 /* Baumgarte 0.2, capped like b2_maxLinearCorrection */
                    if (C < -0.2) C = -0.2;
                    double cx, cy;
                    com(B, &cx, &cy);
                    const double rn = (x - cx) * ny - (y - cy) * nx, k = B->im + B->ii * rn * rn, lam = -C / k;
                    cx += B->im * nx * lam; cy += B->im * ny * lam; B->a += B->ii * rn * lam;
                    const double s2 = sin(B->a), c2 = cos(B->a);
                    B->px = cx - (c2 * B->lcx - s2 * B->lcy); B->py = cy - (s2 * B->lcx + c2 * B->lcy);
                    if (-sep - 3.0 * LIN_SLOP > worst) worst = -sep - 3.0 * LIN_SLOP;
                }
            }
        }
        if (worst <= LIN_SLOP * 0.02) break;
    }
}

static void world_step(L64 *s, double dt)
{
    if (!s->awake) return;
    for (int b = 0; b < NBODY; ++b) {
        Body *B = &s->b[b];
        B->vy += dt * -10.0;
        if (b == 0) { B->vx += dt * B->im * s->fx; B->vy += dt * B->im * s->fy; }
    }
    s->fx = s->fy = 0.0;
    Row rows[MAXROWS];
    int touch_leg[2], touch_hull;
    const int n = build_rows(s, rows, dt, touch_leg, &touch_hull);
    solve_velocity(s, rows, n);
    for (int b = 0; b < NBODY; ++b) {                              /* advance the centres of mass, then the origins */
        Body *B = &s->b[b];
        double cx, cy;
        com(B, &cx, &cy);
        cx += dt * B->vx; cy += dt * B->vy; B->a += dt * B->w;
        const double sn = sin(B->a), c = cos(B->a);
        B->px = cx - (c * B->lcx - sn * B->lcy); B->py = cy - (sn * B->lcx + c * B->lcy);
    }
    solve_position(s);
    /* contact flags of the configuration the step ends in (gym: BeginContact / EndContact listeners) */
    {
        Row scratch[MAXROWS];
        (void)build_rows(s, scratch, dt, touch_leg, &touch_hull);
        s->leg_contact[0] = touch_leg[0]; s->leg_contact[1] = touch_leg[1];
        if (touch_hull) s->game_over = 1;
    }
    int quiet = 1;
    for (int b = 0; b < NBODY; ++b) {
        const Body *B = &s->b[b];
        if (fabs(B->w) > SLEEP_ANG || B->vx * B->vx + B->vy * B->vy > SLEEP_LIN * SLEEP_LIN) quiet = 0;
    }
    s->sleep_time = quiet ? s->sleep_time + dt : 0.0;
    if (s->sleep_time >= SLEEP_TIME) s->awake = 0;
}

static void impulse(L64 *s, double jx, double jy, double px, double py)
{
    Body *B = &s->b[0];
    double cx, cy;
    com(B, &cx, &cy);
    B->vx += B->im * jx; B->vy += B->im * jy;
    B->w += B->ii * ((px - cx) * jy - (py - cy) * jx);
}

void l64_obs(const void *state, double *obs)
{
    const L64 *s = (const L64 *)state;
    const Body *H = &s->b[0];
    /* gym reports lander.linearVelocity: the velocity of the centre of mass */
    obs[0] = (H->px - VW / 2) / (VW / 2);
    obs[1] = (H->py - (s->helipad_y + LEG_DOWN / SCALE)) / (VH / 2);
    obs[2] = H->vx * (VW / 2) / FPS;
    obs[3] = H->vy * (VH / 2) / FPS;
    obs[4] = H->a;
    obs[5] = 20.0 * H->w / FPS;
    obs[6] = s->leg_contact[0] ? 1.0 : 0.0;
    obs[7] = s->leg_contact[1] ? 1.0 : 0.0;
}

/* d0, d1: the two dispersion uniforms in (-1, 1) of this step (inputs: the device draws them from Philox) */
double l64_step(void *state, double a0, double a1, double d0, double d1, int32_t *done)
{
    L64 *s = (L64 *)state;
    d0 /= SCALE; d1 /= SCALE;
    a0 = a0 < -1 ? -1 : (a0 > 1 ? 1 : a0);
    a1 = a1 < -1 ? -1 : (a1 > 1 ? 1 : a1);
    const Body *H = &s->b[0];
    const double tipx = sin(H->a), tipy = cos(H->a), sidex = -tipy, sidey = tipx;
    double m_power = 0, s_power = 0;
    if (a0 > 0.0) {
        m_power = ((a0 < 0 ? 0 : (a0 > 1 ? 1 : a0)) + 1.0) * 0.5;
        const double ox = tipx * (4 / SCALE + 2 * d0) + sidex * d1, oy = -tipy * (4 / SCALE + 2 * d0) - sidey * d1;
        impulse(s, -ox * MAIN_POWER * m_power, -oy * MAIN_POWER * m_power, H->px + ox, H->py + oy);
    }
    if (fabs(a1) > 0.5) {
        const double dir = a1 > 0 ? 1.0 : -1.0;
        s_power = fabs(a1) < 0.5 ? 0.5 : (fabs(a1) > 1 ? 1 : fabs(a1));
        const double ox = tipx * d0 + sidex * (3 * d1 + dir * SIDE_AWAY / SCALE), oy = -tipy * d0 - sidey * (3 * d1 + dir * SIDE_AWAY / SCALE);
        impulse(s, -ox * SIDE_POWER * s_power, -oy * SIDE_POWER * s_power, H->px + ox - tipx * 17 / SCALE, H->py + oy + tipy * SIDE_H / SCALE);
    }
    world_step(s, 1.0 / FPS);
    double o[8];
    l64_obs(s, o);
    const double shaping = -100 * sqrt(o[0] * o[0] + o[1] * o[1]) - 100 * sqrt(o[2] * o[2] + o[3] * o[3]) - 100 * fabs(o[4]) + 10 * o[6] + 10 * o[7];
    double reward = s->has_prev ? shaping - s->prev_shaping : 0.0;
    s->prev_shaping = shaping; s->has_prev = 1;
    reward -= m_power * 0.30;
    reward -= s_power * 0.03;
    *done = 0;
    if (s->game_over || fabs(o[0]) >= 1.0) { *done = 1; reward = -100; }
    if (!s->awake) { *done = 1; reward = 100; }
    return reward;
}

int l64_state_size(void) { return (int)sizeof(L64); }

/* u: the 16 reset uniforms of ses_lander_env.h ([0],[1] initial force, [2..13] terrain chunk heights) */
void l64_reset(void *state, const float *u)
{
    L64 *s = (L64 *)state;
    memset(s, 0, sizeof *s);
    double h[12];
    s->helipad_y = VH / 4;
    for (int i = 0; i < 12; ++i) h[i] = (double)u[2 + i] * (VH / 2);
    for (int i = 3; i <= 7; ++i) h[i] = s->helipad_y;              /* CHUNKS // 2 - 2 .. + 2 */
    for (int i = 0; i < 11; ++i) s->ty[i] = 0.33 * (h[i == 0 ? 11 : i - 1] + h[i] + h[i + 1]);
    static const double hx[6] = {-14, -17, -17, 17, 17, 14}, hy[6] = {17, 0, -10, -10, 0, 17};   /* gym's LANDER_POLY, counter-clockwise */
    double x[6], y[6];
    for (int i = 0; i < 6; ++i) { x[i] = hx[i] / SCALE; y[i] = hy[i] / SCALE; }
    set_polygon(&s->b[0], 6, x, y, 5.0, 0.1);
    const double lx[4] = {-LEG_W / SCALE, LEG_W / SCALE, LEG_W / SCALE, -LEG_W / SCALE}, ly[4] = {-LEG_H / SCALE, -LEG_H / SCALE, LEG_H / SCALE, LEG_H / SCALE};
    for (int leg = 0; leg < 2; ++leg) {
        set_polygon(&s->b[1 + leg], 4, lx, ly, 1.0, 0.2);           /* b2FixtureDef default friction 0.2 */
        const double i = leg == 0 ? -1.0 : 1.0;
        s->b[1 + leg].px = VW / 2 - i * LEG_AWAY / SCALE; s->b[1 + leg].py = VH; s->b[1 + leg].a = i * 0.05;
    }
    s->b[0].px = VW / 2; s->b[0].py = VH;
    s->fx = 2000.0 * (double)u[0] - 1000.0; s->fy = 2000.0 * (double)u[1] - 1000.0;
    s->awake = 1;
    int32_t done;
    (void)l64_step(s, 0.0, 0.0, 0.0, 0.0, &done);                    /* gym's reset ends with step(noop); no engine fires: no noise used */
}

/* Adopt the configuration another integration is in after ITS reset: bodies[3][6] = centre of mass x, y, angle, velocity
 * of the centre of mass, angular velocity (what o_lander_debug reports).  gym creates the legs 0.6 m off their hip anchors
 * and lets the first world.Step pull them in; where that snap leaves the legs inside their limits depends on the path
 * Box2D's position solver takes (maxLinearCorrection per iteration, block order) -- an artefact of that solver, not
 * physics an independent integration can be expected to reproduce.  The comparison therefore starts after it. */
void l64_adopt(void *state, const double *bodies)
{
    L64 *s = (L64 *)state;
    for (int b = 0; b < NBODY; ++b) {
        Body *B = &s->b[b];
        const double *v = bodies + 6 * b;
        B->a = v[2];
        const double sn = sin(B->a), c = cos(B->a);
        B->px = v[0] - (c * B->lcx - sn * B->lcy); B->py = v[1] - (sn * B->lcx + c * B->lcy);
        B->vx = v[3]; B->vy = v[4]; B->w = v[5];
    }
    s->fx = s->fy = 0.0;
    s->sleep_time = 0.0; s->awake = 1; s->game_over = 0; s->leg_contact[0] = s->leg_contact[1] = 0;
    double o[8];
    l64_obs(s, o);
    s->prev_shaping = -100 * sqrt(o[0] * o[0] + o[1] * o[1]) - 100 * sqrt(o[2] * o[2] + o[3] * o[3]) - 100 * fabs(o[4]) + 10 * o[6] + 10 * o[7];
    s->has_prev = 1;
}

void l64_debug(const void *state, double *bodies /* [3][6] origin x, y, angle, v of the centre of mass, w */, int32_t *flags)
{
    const L64 *s = (const L64 *)state;
    for (int b = 0; b < NBODY; ++b) {
        const Body *B = &s->b[b];
        const double v[6] = {B->px, B->py, B->a, B->vx, B->vy, B->w};
        memcpy(bodies + 6 * b, v, sizeof v);
    }
    flags[0] = s->game_over; flags[1] = s->awake; flags[2] = s->leg_contact[0]; flags[3] = s->leg_contact[1];
}

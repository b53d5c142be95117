// ses_b2_toi.h -- continuous collision between a moving body polygon and a static terrain edge, in the manner of
// Box2D 2.3.0: b2Distance (GJK with a cached simplex), b2SeparationFunction and b2TimeOfImpact (conservative
// advancement with a bracketed root finder).  Included by ses_b2.h; the world-level sub-stepping (b2World::SolveTOI,
// b2Island::SolveTOI) is there.
//
// One text compiled twice, like ses_b2.h (see there): for gfx950 by the product, for the host by oracle/ses_b2_oracle.cpp
// (test infrastructure, -I simple-es_amd/csrc).  PARITY WITH BOX2D IS UNPINNED (Box2D is not in the reference tree nor in
// this image): what follows restates the published algorithms of b2Distance.cpp and b2TimeOfImpact.cpp for the one pair
// of shapes these worlds have -- proxy A = a terrain edge (two vertices, in world coordinates: the terrain body's
// transform is the identity and it never moves), proxy B = a body's convex polygon (<= 6 vertices, swept from
// (c0, a0) to (c, a) about its local centre) -- with both radii = b2_polygonRadius.
#pragma once

namespace b2l {

struct Sweep {                       // b2Sweep of one body during a step
    float c0x, c0y, a0;              // centre of mass / angle at time alpha0
    float cx, cy, a;                 // ... at time 1
    float alpha0;
};

// b2Sweep::GetTransform
B2_FN void sweep_xf(const Sweep &s, const BodyDef &d, float beta, Xf &x)
{
    const float px = (1.0f - beta) * s.c0x + beta * s.cx, py = (1.0f - beta) * s.c0y + beta * s.cy;
    const float angle = (1.0f - beta) * s.a0 + beta * s.a;
    float sn, cs;
    B2_SINCOS(angle, sn, cs);
    x.s = sn; x.c = cs;
    x.px = px - (cs * d.lcx - sn * d.lcy);
    x.py = py - (sn * d.lcx + cs * d.lcy);
}

// b2Sweep::Advance
B2_FN void sweep_advance(Sweep &s, float alpha)
{
    const float beta = (alpha - s.alpha0) / (1.0f - s.alpha0);
    s.c0x += beta * (s.cx - s.c0x);
    s.c0y += beta * (s.cy - s.c0y);
    s.a0 += beta * (s.a - s.a0);
    s.alpha0 = alpha;
}

struct ToiPair {                     // the two b2DistanceProxy of a (terrain edge, body polygon) pair
    float ex[2], ey[2];              // proxy A: the edge's vertices (world = local)
    const Poly *P;                   // proxy B: local vertices P->vx[i], P->vy[i], i < P->n
};

// What time_of_impact works on: the pair by value, polygon included.  Vertices are picked by index with selects (vertex_x /
// vertex_y) and every loop over them has a constant trip count, so on the device the whole pair sits in registers -- read
// through the ToiPair it was a load from (constant or private) memory per vertex access inside the root finder's loops.
struct ToiLocal {
    float ex[2], ey[2];
    float vx[6], vy[6];
    int n;
};

B2_FN float edge_x(const ToiLocal &p, int i) { return i ? p.ex[1] : p.ex[0]; }    // selects, not indexed private memory
B2_FN float edge_y(const ToiLocal &p, int i) { return i ? p.ey[1] : p.ey[0]; }
B2_FN float vertex_x(const ToiLocal &p, int i)
{
    float r = p.vx[0];
    B2_UNROLL
    for (int k = 1; k < 6; ++k) r = i == k ? p.vx[k] : r;
    return r;
}
B2_FN float vertex_y(const ToiLocal &p, int i)
{
    float r = p.vy[0];
    B2_UNROLL
    for (int k = 1; k < 6; ++k) r = i == k ? p.vy[k] : r;
    return r;
}

// b2DistanceProxy::GetSupport
B2_FN int support_edge(const ToiLocal &p, float dx, float dy)
{
    const float v0 = p.ex[0] * dx + p.ey[0] * dy, v1 = p.ex[1] * dx + p.ey[1] * dy;
    return v1 > v0 ? 1 : 0;
}
B2_FN int support_poly(const ToiLocal &p, float dx, float dy)
{
    int best = 0;
    float best_value = p.vx[0] * dx + p.vy[0] * dy;
    B2_UNROLL
    for (int i = 1; i < 6; ++i) {
        if (i < p.n) {
            const float value = p.vx[i] * dx + p.vy[i] * dy;
            if (value > best_value) { best = i; best_value = value; }
        }
    }
    return best;
}

struct SimplexCache {                // b2SimplexCache
    float metric;
    int count;
    int ia[3], ib[3];
};

struct SVertex {                     // b2SimplexVertex
    float wax, way, wbx, wby, wx, wy, a;
    int ia, ib;
};

B2_FN void svertex_set(SVertex &v, const ToiLocal &p, const Xf &xfB, int ia, int ib)
{
    v.ia = ia; v.ib = ib;
    v.wax = edge_x(p, ia); v.way = edge_y(p, ia);                              // b2Mul(identity, vertexA)
    const float lx = vertex_x(p, ib), ly = vertex_y(p, ib);
    v.wbx = (xfB.c * lx - xfB.s * ly) + xfB.px; v.wby = (xfB.s * lx + xfB.c * ly) + xfB.py;
    v.wx = v.wbx - v.wax; v.wy = v.wby - v.way;
}

B2_FN float simplex_metric(const SVertex (&v)[3], int count)
{
    if (count == 2) {
        const float dx = v[0].wx - v[1].wx, dy = v[0].wy - v[1].wy;
        return B2_SQRT(dx * dx + dy * dy);
    }
    if (count == 3) {
        const float ax = v[1].wx - v[0].wx, ay = v[1].wy - v[0].wy, bx = v[2].wx - v[0].wx, by = v[2].wy - v[0].wy;
        return ax * by - ay * bx;
    }
    return 0.0f;
}

// b2Distance (useRadii = false): distance between the core shapes at transform xfB of the polygon; updates the cache
B2_FN float gjk_distance(SimplexCache &cache, const ToiLocal &p, const Xf &xfB)
{
    constexpr float EPS = 1.1920928955078125e-7f;
    SVertex v[3];
    int count = cache.count;
    // b2Simplex::ReadCache
    for (int i = 0; i < 3; ++i) {
        if (i < count) { svertex_set(v[i], p, xfB, cache.ia[i], cache.ib[i]); v[i].a = 0.0f; }
    }
    if (count > 1) {
        const float metric1 = cache.metric, metric2 = simplex_metric(v, count);
        if (metric2 < 0.5f * metric1 || 2.0f * metric1 < metric2 || metric2 < EPS) count = 0;
    }
    if (count == 0) {
        svertex_set(v[0], p, xfB, 0, 0);
        v[0].a = 1.0f;
        count = 1;
    }
    int save_a[3], save_b[3];
    for (int iter = 0; iter < 20;) {
        const int save_count = count;
        for (int i = 0; i < 3; ++i) {
            if (i < save_count) { save_a[i] = v[i].ia; save_b[i] = v[i].ib; }
        }
        if (count == 2) {                                                    // b2Simplex::Solve2
            const float e12x = v[1].wx - v[0].wx, e12y = v[1].wy - v[0].wy;
            const float d12_2 = -(v[0].wx * e12x + v[0].wy * e12y);
            const float d12_1 = v[1].wx * e12x + v[1].wy * e12y;
            if (d12_2 <= 0.0f) {
                v[0].a = 1.0f; count = 1;
            } else if (d12_1 <= 0.0f) {
                v[1].a = 1.0f; count = 1; v[0] = v[1];
            } else {
                const float inv = 1.0f / (d12_1 + d12_2);
                v[0].a = d12_1 * inv; v[1].a = d12_2 * inv;
            }
        } else if (count == 3) {                                             // b2Simplex::Solve3
            const float w1x = v[0].wx, w1y = v[0].wy, w2x = v[1].wx, w2y = v[1].wy, w3x = v[2].wx, w3y = v[2].wy;
            const float e12x = w2x - w1x, e12y = w2y - w1y;
            const float d12_1 = w2x * e12x + w2y * e12y, d12_2 = -(w1x * e12x + w1y * e12y);
            const float e13x = w3x - w1x, e13y = w3y - w1y;
            const float d13_1 = w3x * e13x + w3y * e13y, d13_2 = -(w1x * e13x + w1y * e13y);
            const float e23x = w3x - w2x, e23y = w3y - w2y;
            const float d23_1 = w3x * e23x + w3y * e23y, d23_2 = -(w2x * e23x + w2y * e23y);
            const float n123 = e12x * e13y - e12y * e13x;
            const float d123_1 = n123 * (w2x * w3y - w2y * w3x), d123_2 = n123 * (w3x * w1y - w3y * w1x),
                        d123_3 = n123 * (w1x * w2y - w1y * w2x);
            if (d12_2 <= 0.0f && d13_2 <= 0.0f) {
                v[0].a = 1.0f; count = 1;
            } else if (d12_1 > 0.0f && d12_2 > 0.0f && d123_3 <= 0.0f) {
                const float inv = 1.0f / (d12_1 + d12_2);
                v[0].a = d12_1 * inv; v[1].a = d12_2 * inv; count = 2;
            } else if (d13_1 > 0.0f && d13_2 > 0.0f && d123_2 <= 0.0f) {
                const float inv = 1.0f / (d13_1 + d13_2);
                v[0].a = d13_1 * inv; v[2].a = d13_2 * inv; count = 2; v[1] = v[2];
            } else if (d12_1 <= 0.0f && d23_2 <= 0.0f) {
                v[1].a = 1.0f; count = 1; v[0] = v[1];
            } else if (d13_1 <= 0.0f && d23_1 <= 0.0f) {
                v[2].a = 1.0f; count = 1; v[0] = v[2];
            } else if (d23_1 > 0.0f && d23_2 > 0.0f && d123_1 <= 0.0f) {
                const float inv = 1.0f / (d23_1 + d23_2);
                v[1].a = d23_1 * inv; v[2].a = d23_2 * inv; count = 2; v[0] = v[2];
            } else {
                const float inv = 1.0f / (d123_1 + d123_2 + d123_3);
                v[0].a = d123_1 * inv; v[1].a = d123_2 * inv; v[2].a = d123_3 * inv; count = 3;
            }
        }
        if (count == 3) break;                                               // the origin is inside: overlap
        // b2Simplex::GetSearchDirection
        float dx, dy;
        if (count == 1) {
            dx = -v[0].wx; dy = -v[0].wy;
        } else {
            const float e12x = v[1].wx - v[0].wx, e12y = v[1].wy - v[0].wy;
            const float sgn = e12x * (-v[0].wy) - e12y * (-v[0].wx);
            if (sgn > 0.0f) { dx = -1.0f * e12y; dy = 1.0f * e12x; }        // b2Cross(1.0f, e12)
            else { dx = 1.0f * e12y; dy = -1.0f * e12x; }                    // b2Cross(e12, 1.0f)
        }
        if (dx * dx + dy * dy < EPS * EPS) break;
        // new vertex from the support points along -d (A) and d (B)
        SVertex nv;
        const int ia = support_edge(p, -dx, -dy);
        const int ib = support_poly(p, xfB.c * dx + xfB.s * dy, -xfB.s * dx + xfB.c * dy);      // b2MulT(xfB.q, d)
        svertex_set(nv, p, xfB, ia, ib);
        nv.a = 0.0f;
        if (count == 1) v[1] = nv; else v[2] = nv;                           // vertices + m_count
        ++iter;
        bool duplicate = false;
        for (int i = 0; i < 3; ++i) {
            if (i < save_count && ia == save_a[i] && ib == save_b[i]) duplicate = true;
        }
        if (duplicate) break;
        ++count;
    }
    // witness points and distance
    float pax, pay, pbx, pby;
    if (count == 1) {
        pax = v[0].wax; pay = v[0].way; pbx = v[0].wbx; pby = v[0].wby;
    } else if (count == 2) {
        pax = v[0].a * v[0].wax + v[1].a * v[1].wax; pay = v[0].a * v[0].way + v[1].a * v[1].way;
        pbx = v[0].a * v[0].wbx + v[1].a * v[1].wbx; pby = v[0].a * v[0].wby + v[1].a * v[1].wby;
    } else {
        pax = v[0].a * v[0].wax + v[1].a * v[1].wax + v[2].a * v[2].wax;
        pay = v[0].a * v[0].way + v[1].a * v[1].way + v[2].a * v[2].way;
        pbx = pax; pby = pay;
    }
    // b2Simplex::WriteCache
    cache.metric = simplex_metric(v, count);
    cache.count = count;
    for (int i = 0; i < 3; ++i) {
        if (i < count) { cache.ia[i] = v[i].ia; cache.ib[i] = v[i].ib; }
    }
    const float ddx = pax - pbx, ddy = pay - pby;
    return B2_SQRT(ddx * ddx + ddy * ddy);
}

// b2SeparationFunction
struct SepFn {
    int type;                        // 0 = e_points, 1 = e_faceA, 2 = e_faceB
    float lpx, lpy;                  // m_localPoint
    float ax, ay;                    // m_axis
};

B2_FN void sepfn_init(SepFn &f, const SimplexCache &cache, const ToiLocal &p, const Sweep &sw, const BodyDef &bd, float t1)
{
    Xf xfB;
    sweep_xf(sw, bd, t1, xfB);
    if (cache.count == 1) {
        f.type = 0;
        const float pax = edge_x(p, cache.ia[0]), pay = edge_y(p, cache.ia[0]);
        const float lx = vertex_x(p, cache.ib[0]), ly = vertex_y(p, cache.ib[0]);
        const float pbx = (xfB.c * lx - xfB.s * ly) + xfB.px, pby = (xfB.s * lx + xfB.c * ly) + xfB.py;
        f.ax = pbx - pax; f.ay = pby - pay;
        const float len = B2_SQRT(f.ax * f.ax + f.ay * f.ay);               // b2Vec2::Normalize
        if (!(len < 1.1920928955078125e-7f)) { const float inv = 1.0f / len; f.ax *= inv; f.ay *= inv; }
        f.lpx = 0.0f; f.lpy = 0.0f;
    } else if (cache.ia[0] == cache.ia[1]) {                                 // two points on B, one on A
        f.type = 2;
        const float b1x = vertex_x(p, cache.ib[0]), b1y = vertex_y(p, cache.ib[0]), b2x = vertex_x(p, cache.ib[1]), b2y = vertex_y(p, cache.ib[1]);
        f.ax = 1.0f * (b2y - b1y); f.ay = -1.0f * (b2x - b1x);              // b2Cross(b2 - b1, 1.0f)
        const float len = B2_SQRT(f.ax * f.ax + f.ay * f.ay);
        if (!(len < 1.1920928955078125e-7f)) { const float inv = 1.0f / len; f.ax *= inv; f.ay *= inv; }
        const float nx = xfB.c * f.ax - xfB.s * f.ay, ny = xfB.s * f.ax + xfB.c * f.ay;
        f.lpx = 0.5f * (b1x + b2x); f.lpy = 0.5f * (b1y + b2y);
        const float pbx = (xfB.c * f.lpx - xfB.s * f.lpy) + xfB.px, pby = (xfB.s * f.lpx + xfB.c * f.lpy) + xfB.py;
        const float pax = edge_x(p, cache.ia[0]), pay = edge_y(p, cache.ia[0]);
        const float s = (pax - pbx) * nx + (pay - pby) * ny;
        if (s < 0.0f) { f.ax = -f.ax; f.ay = -f.ay; }
    } else {                                                                 // two points on A, one or two on B
        f.type = 1;
        const float a1x = edge_x(p, cache.ia[0]), a1y = edge_y(p, cache.ia[0]), a2x = edge_x(p, cache.ia[1]), a2y = edge_y(p, cache.ia[1]);
        f.ax = 1.0f * (a2y - a1y); f.ay = -1.0f * (a2x - a1x);
        const float len = B2_SQRT(f.ax * f.ax + f.ay * f.ay);
        if (!(len < 1.1920928955078125e-7f)) { const float inv = 1.0f / len; f.ax *= inv; f.ay *= inv; }
        f.lpx = 0.5f * (a1x + a2x); f.lpy = 0.5f * (a1y + a2y);
        const float lx = vertex_x(p, cache.ib[0]), ly = vertex_y(p, cache.ib[0]);
        const float pbx = (xfB.c * lx - xfB.s * ly) + xfB.px, pby = (xfB.s * lx + xfB.c * ly) + xfB.py;
        const float s = (pbx - f.lpx) * f.ax + (pby - f.lpy) * f.ay;
        if (s < 0.0f) { f.ax = -f.ax; f.ay = -f.ay; }
    }
}

// b2SeparationFunction::FindMinSeparation (find == true: the support indices are chosen and returned) and ::Evaluate
B2_FN float sepfn_eval(const SepFn &f, const ToiLocal &p, const Sweep &sw, const BodyDef &bd, float t, bool find, int &ia, int &ib)
{
    Xf xfB;
    sweep_xf(sw, bd, t, xfB);
    if (f.type == 0) {
        if (find) {
            ia = support_edge(p, f.ax, f.ay);
            ib = support_poly(p, xfB.c * (-f.ax) + xfB.s * (-f.ay), -xfB.s * (-f.ax) + xfB.c * (-f.ay));
        }
        const float lx = vertex_x(p, ib), ly = vertex_y(p, ib);
        const float pbx = (xfB.c * lx - xfB.s * ly) + xfB.px, pby = (xfB.s * lx + xfB.c * ly) + xfB.py;
        return (pbx - edge_x(p, ia)) * f.ax + (pby - edge_y(p, ia)) * f.ay;
    }
    if (f.type == 1) {
        if (find) {
            ia = -1;
            ib = support_poly(p, xfB.c * (-f.ax) + xfB.s * (-f.ay), -xfB.s * (-f.ax) + xfB.c * (-f.ay));
        }
        const float lx = vertex_x(p, ib), ly = vertex_y(p, ib);
        const float pbx = (xfB.c * lx - xfB.s * ly) + xfB.px, pby = (xfB.s * lx + xfB.c * ly) + xfB.py;
        return (pbx - f.lpx) * f.ax + (pby - f.lpy) * f.ay;
    }
    const float nx = xfB.c * f.ax - xfB.s * f.ay, ny = xfB.s * f.ax + xfB.c * f.ay;
    const float pbx = (xfB.c * f.lpx - xfB.s * f.lpy) + xfB.px, pby = (xfB.s * f.lpx + xfB.c * f.lpy) + xfB.py;
    if (find) {
        ib = -1;
        ia = support_edge(p, -nx, -ny);
    }
    return (edge_x(p, ia) - pbx) * nx + (edge_y(p, ia) - pby) * ny;
}

constexpr int TOI_FAILED = 0, TOI_OVERLAPPED = 1, TOI_TOUCHING = 2, TOI_SEPARATED = 3;

// b2TimeOfImpact with tMax = 1: the fraction t of the sweep at which the core shapes are `target` apart
// A REAL function (B2_NOINLINE): called once per (body, candidate edge) and step; inlined at every call site it would
// multiply the world step's code
B2_NOINLINE int time_of_impact(const ToiPair &pair, const Sweep &sweep, const BodyDef &body_def, float &t_result)
{
    // everything the iterations read, by value (see ToiLocal); the result leaves through t_result once, at the end
    ToiLocal p;
    p.ex[0] = pair.ex[0]; p.ex[1] = pair.ex[1]; p.ey[0] = pair.ey[0]; p.ey[1] = pair.ey[1];
    p.n = pair.P->n;
    B2_UNROLL
    for (int i = 0; i < 6; ++i) { p.vx[i] = pair.P->vx[i]; p.vy[i] = pair.P->vy[i]; }
    const Sweep sw = sweep;
    const BodyDef bd = body_def;
    float t_out;
    const float total_radius = POLY_RADIUS + POLY_RADIUS;
    const float target = b2max(LINEAR_SLOP, total_radius - 3.0f * LINEAR_SLOP);
    const float tolerance = 0.25f * LINEAR_SLOP;
    const float t_max = 1.0f;
    float t1 = 0.0f;
    int state = TOI_FAILED;
    t_out = t_max;
    SimplexCache cache;
    cache.count = 0; cache.metric = 0.0f;
    for (int i = 0; i < 3; ++i) { cache.ia[i] = 0; cache.ib[i] = 0; }
    for (int iter = 0;;) {
        Xf xfB;
        sweep_xf(sw, bd, t1, xfB);
        const float distance = gjk_distance(cache, p, xfB);
        if (distance <= 0.0f) { state = TOI_OVERLAPPED; t_out = 0.0f; break; }
        if (distance < target + tolerance) { state = TOI_TOUCHING; t_out = t1; break; }
        SepFn fcn;
        sepfn_init(fcn, cache, p, sw, bd, t1);
        bool done = false;
        float t2 = t_max;
        for (int push_back = 0;;) {
            int ia = 0, ib = 0;
            float s2 = sepfn_eval(fcn, p, sw, bd, t2, true, ia, ib);
            if (s2 > target + tolerance) { state = TOI_SEPARATED; t_out = t_max; done = true; break; }
            if (s2 > target - tolerance) { t1 = t2; break; }
            float s1 = sepfn_eval(fcn, p, sw, bd, t1, false, ia, ib);
            if (s1 < target - tolerance) { state = TOI_FAILED; t_out = t1; done = true; break; }
            if (s1 <= target + tolerance) { state = TOI_TOUCHING; t_out = t1; done = true; break; }
            float a1 = t1, a2 = t2;
            for (int root = 0;;) {
                float t;
                if (root & 1) t = a1 + (target - s1) * (a2 - a1) / (s2 - s1);
                else t = 0.5f * (a1 + a2);
                ++root;
                const float s = sepfn_eval(fcn, p, sw, bd, t, false, ia, ib);
                if (b2abs(s - target) < tolerance) { t2 = t; break; }
                if (s > target) { a1 = t; s1 = s; } else { a2 = t; s2 = s; }
                if (root == 50) break;
            }
            ++push_back;
            if (push_back == 8) break;                                       // b2_maxPolygonVertices
        }
        ++iter;
        if (done) break;
        if (iter == 20) { state = TOI_FAILED; t_out = t1; break; }
    }
    t_result = t_out;
    return state;
}

}  // namespace b2l

#!/usr/bin/env python3
"""Generates simple-es_amd/csrc/ses_b2_shapes.h (the oracle compiles the same file, oracle/Makefile): the
constant polygon and mass tables of the rigid-body worlds behind LunarLanderContinuous-v2 and BipedalWalker-v3.

gym builds these bodies through pybox2d (Box2D 2.3.0): `polygonShape(vertices=...)` runs b2PolygonShape::Set (convex
hull ordering from the right-most vertex, counter-clockwise; edge normals; centroid), fixture creation runs
b2PolygonShape::ComputeMass and b2Body::ResetMassData.  Box2D computes all of this in float32 at body creation; the
results never change, so they are tabulated here with numpy float32 arithmetic in Box2D's operation order and emitted
as hexadecimal float literals (exact)."""
import os
import sys

import numpy as np

f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hull_order(pts):
    """b2PolygonShape::Set: gift wrapping from the right-most (lowest on ties) point, counter-clockwise."""
    ps = [(f32(x), f32(y)) for x, y in pts]
    n = len(ps)
    i0 = 0
    x0 = ps[0][0]
    for i in range(1, n):
        x = ps[i][0]
        if x > x0 or (x == x0 and ps[i][1] < ps[i0][1]):
            i0, x0 = i, x
    hull, ih = [], i0
    while True:
        hull.append(ih)
        ie = 0
        for j in range(1, n):
            if ie == ih:
                ie = j
                continue
            rx, ry = ps[ie][0] - ps[hull[-1]][0], ps[ie][1] - ps[hull[-1]][1]
            vx, vy = ps[j][0] - ps[hull[-1]][0], ps[j][1] - ps[hull[-1]][1]
            c = f32(rx * vy) - f32(ry * vx)
            if c < 0:
                ie = j
            if c == 0 and f32(vx * vx + vy * vy) > f32(rx * rx + ry * ry):
                ie = j
        ih = ie
        if ie == i0:
            break
    return [ps[i] for i in hull]


def normals_of(vs):
    out = []
    n = len(vs)
    for i in range(n):
        i2 = (i + 1) % n
        ex, ey = f32(vs[i2][0] - vs[i][0]), f32(vs[i2][1] - vs[i][1])
        nx, ny = f32(f32(1.0) * ey), f32(f32(-1.0) * ex)            # b2Cross(edge, 1.0f)
        length = f32(np.sqrt(f32(f32(nx * nx) + f32(ny * ny))))
        inv = f32(f32(1.0) / length)
        out.append((f32(nx * inv), f32(ny * inv)))
    return out


def centroid_of(vs):
    """b2PolygonShape.cpp ComputeCentroid (reference point = origin)."""
    cx = cy = area = f32(0)
    inv3 = f32(f32(1.0) / f32(3.0))
    n = len(vs)
    for i in range(n):
        p2, p3 = vs[i], vs[(i + 1) % n]
        e1x, e1y, e2x, e2y = p2[0], p2[1], p3[0], p3[1]
        D = f32(f32(e1x * e2y) - f32(e1y * e2x))
        ta = f32(f32(0.5) * D)
        area = f32(area + ta)
        k = f32(ta * inv3)
        cx = f32(cx + f32(k * f32(f32(f32(0) + p2[0]) + p3[0])))
        cy = f32(cy + f32(k * f32(f32(f32(0) + p2[1]) + p3[1])))
    inv = f32(f32(1.0) / area)
    return f32(cx * inv), f32(cy * inv)


def mass_of(vs, density):
    """b2PolygonShape::ComputeMass (2.3.0: reference point s = vertex average) -> mass, centre, I about the origin."""
    n = len(vs)
    sx = sy = f32(0)
    for v in vs:
        sx, sy = f32(sx + v[0]), f32(sy + v[1])
    k = f32(f32(1.0) / f32(n))
    sx, sy = f32(sx * k), f32(sy * k)
    cx = cy = area = inertia = f32(0)
    inv3 = f32(f32(1.0) / f32(3.0))
    for i in range(n):
        e1x, e1y = f32(vs[i][0] - sx), f32(vs[i][1] - sy)
        j = (i + 1) % n
        e2x, e2y = f32(vs[j][0] - sx), f32(vs[j][1] - sy)
        D = f32(f32(e1x * e2y) - f32(e1y * e2x))
        ta = f32(f32(0.5) * D)
        area = f32(area + ta)
        kk = f32(ta * inv3)
        cx = f32(cx + f32(kk * f32(e1x + e2x)))
        cy = f32(cy + f32(kk * f32(e1y + e2y)))
        intx2 = f32(f32(f32(e1x * e1x) + f32(e2x * e1x)) + f32(e2x * e2x))
        inty2 = f32(f32(f32(e1y * e1y) + f32(e2y * e1y)) + f32(e2y * e2y))
        inertia = f32(inertia + f32(f32(f32(f32(0.25) * inv3) * D) * f32(intx2 + inty2)))
    mass = f32(f32(density) * area)
    inv = f32(f32(1.0) / area)
    cx, cy = f32(cx * inv), f32(cy * inv)
    mcx, mcy = f32(cx + sx), f32(cy + sy)
    I = f32(f32(density) * inertia)
    I = f32(I + f32(mass * f32(f32(f32(mcx * mcx) + f32(mcy * mcy)) - f32(f32(cx * cx) + f32(cy * cy)))))
    return mass, (mcx, mcy), I


def body_of(vs, density):
    """b2Body::ResetMassData for a body with this single fixture."""
    mass, (cx, cy), I = mass_of(vs, density)
    inv_mass = f32(f32(1.0) / mass)
    lcx, lcy = f32(f32(mass * cx) * inv_mass), f32(f32(mass * cy) * inv_mass)
    I = f32(I - f32(mass * f32(f32(lcx * lcx) + f32(lcy * lcy))))
    return {"mass": mass, "inv_mass": inv_mass, "inv_i": f32(f32(1.0) / I), "lc": (lcx, lcy)}


def box(hx, hy):
    hx, hy = f32(hx), f32(hy)
    return [(-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy)], [(f32(0), f32(-1)), (f32(1), f32(0)), (f32(0), f32(1)), (f32(-1), f32(0))], (f32(0), f32(0))


def h(x):
    return float(f32(x)).hex() + "f"


def emit_poly(name, vs, ns, c, maxv):
    pad = lambda seq: list(seq) + [f32(0)] * (maxv - len(seq))
    rows = ["{%s}" % ", ".join(h(v) for v in pad([p[0] for p in vs])), "{%s}" % ", ".join(h(v) for v in pad([p[1] for p in vs])),
            "{%s}" % ", ".join(h(v) for v in pad([p[0] for p in ns])), "{%s}" % ", ".join(h(v) for v in pad([p[1] for p in ns]))]
    return "    /* %s */ {%d, %s, %s, %s}" % (name, len(vs), ",\n        ".join(rows), h(c[0]), h(c[1]))


def emit_body(name, b, friction):
    return "    /* %s: mass %.7g */ {%s, %s, %s, %s, %s}" % (name, float(b["mass"]), h(b["inv_mass"]), h(b["inv_i"]),
                                                          h(b["lc"][0]), h(b["lc"][1]), h(friction))


def mix_friction(a, b):
    return f32(np.sqrt(f32(f32(a) * f32(b))))


def main():
    SCALE = 30.0
    out = ["// GENERATED by tools/gen_b2_shapes.py -- do not edit.  Polygon / mass tables of the LunarLander and",
           "// BipedalWalker bodies as Box2D 2.3.0 computes them at body creation (float32, Box2D's operation order).",
           "#pragma once", "", "namespace b2l {", "",
           "struct Poly {", "    int n;", "    float vx[6], vy[6], nx[6], ny[6];", "    float cx, cy;   // centroid", "};",
           "struct BodyDef {", "    float inv_mass, inv_i, lcx, lcy, friction;   // friction: mixed with the terrain's, sqrt(f1 * f2)",
           "};", ""]

    # ---- LunarLander (gym lunar_lander.py) ----
    hull = hull_order([(x / SCALE, y / SCALE) for x, y in [(-14, +17), (-17, 0), (-17, -10), (+17, -10), (+17, 0), (+14, +17)]])
    hn, hc = normals_of(hull), centroid_of(hull)
    leg, ln, lc = box(2 / SCALE, 8 / SCALE)
    out.append("B2_CONST Poly LANDER_POLY[3] = {")
    out.append(",\n".join([emit_poly("lander hull", hull, hn, hc, 6), emit_poly("leg -1", leg, ln, lc, 6),
                           emit_poly("leg +1", leg, ln, lc, 6)]))
    out.append("};")
    out.append("B2_CONST BodyDef LANDER_BODY[3] = {")
    out.append(",\n".join([emit_body("lander hull, density 5", body_of(hull, 5.0), mix_friction(0.1, 0.1)),
                           emit_body("leg, density 1", body_of(leg, 1.0), mix_friction(0.1, 0.2)),
                           emit_body("leg, density 1", body_of(leg, 1.0), mix_friction(0.1, 0.2))]))
    out.append("};")
    out.append("")

    # ---- BipedalWalker (gym bipedal_walker.py) ----
    whull = hull_order([(x / SCALE, y / SCALE) for x, y in [(-30, +9), (+6, +9), (+34, +1), (+34, -8), (-30, -8)]])
    wn, wc = normals_of(whull), centroid_of(whull)
    LEG_W, LEG_H = 8 / SCALE, 34 / SCALE
    up, upn, upc = box(LEG_W / 2, LEG_H / 2)
    lo, lon, loc = box(0.8 * LEG_W / 2, LEG_H / 2)
    out.append("B2_CONST Poly WALKER_POLY[5] = {")
    out.append(",\n".join([emit_poly("walker hull", whull, wn, wc, 6), emit_poly("upper leg -1", up, upn, upc, 6),
                           emit_poly("lower leg -1", lo, lon, loc, 6), emit_poly("upper leg +1", up, upn, upc, 6),
                           emit_poly("lower leg +1", lo, lon, loc, 6)]))
    out.append("};")
    out.append("B2_CONST BodyDef WALKER_BODY[5] = {")
    fr = mix_friction(2.5, 0.2)                       # terrain FRICTION 2.5, fixtureDef default 0.2
    out.append(",\n".join([emit_body("walker hull, density 5", body_of(whull, 5.0), mix_friction(2.5, 0.1)),
                           emit_body("upper leg, density 1", body_of(up, 1.0), fr),
                           emit_body("lower leg, density 1", body_of(lo, 1.0), fr),
                           emit_body("upper leg, density 1", body_of(up, 1.0), fr),
                           emit_body("lower leg, density 1", body_of(lo, 1.0), fr)]))
    out.append("};")
    out.append("")
    out.append("}  // namespace b2l")
    text = "\n".join(out) + "\n"
    for rel in ("simple-es_amd/csrc/ses_b2_shapes.h",):
        with open(os.path.join(ROOT, rel), "w") as fh:
            fh.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()

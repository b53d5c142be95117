"""BipedalWalker-v3 on the CPU (ses_walker_env.h over the Box2D-style world of ses_b2.h, compiled for the host): behavioural
checks, and the envelope around an independently written float64 integration (oracle/walker64.c) -- gym and Box2D are
not here to compare with (parity unpinned, see the headers)."""
import os

import numpy as np

from oracle import c_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_walker_header_exists_once():
    # one text, compiled for gfx950 by the product and for the host by the oracle (oracle/Makefile: -I simple-es_amd/csrc)
    assert os.path.exists(os.path.join(ROOT, "simple-es_amd", "csrc", "ses_walker_env.h"))
    assert not os.path.exists(os.path.join(ROOT, "oracle", "ses_walker_env.h"))


def test_terrain_is_gyms_random_walk():
    sim = co.WalkerSim()
    sim.reset(np.array([0.5, 0.123, 0.456, 0.0], np.float32))
    _, terr, _ = sim.debug()
    th = 400 / 30 / 4
    assert np.allclose(terr[:21], th)                              # TERRAIN_STARTPAD: flat start
    assert np.abs(np.diff(terr)).max() < 0.2 and abs(terr.mean() - th) < 0.5
    assert np.unique(terr[21:]).size > 120                         # a random walk with a flat repeat every 5..9 points
    sim.reset(np.array([0.5, 0.124, 0.456, 0.0], np.float32))
    assert not np.array_equal(sim.debug()[1], terr)                # keyed by the init row


def _run(sim, u, ctrl, limit=300):
    obs = sim.reset(u)
    tot, t, done = 0.0, 0, False
    while not done and t < limit:
        obs, r, done = sim.step(ctrl(obs, t))
        tot += r
        t += 1
    return tot, t, obs


def test_walker_behaves_like_gyms():
    rng = np.random.RandomState(0)
    sim = co.WalkerSim()
    obs = sim.reset(rng.rand(4).astype(np.float32))
    assert obs.shape == (24,) and abs(obs[0]) < 0.05               # upright hull after reset
    lidar = obs[14:]
    assert np.all(np.diff(lidar[:8]) > 0) and lidar[0] > 0.3 and lidar[9] == 1.0   # rays fan out from straight down
    # no torque: the legs fold, the hull hits the ground: -100 (gym: same, after ~100 steps)
    tot, t, _ = _run(sim, rng.rand(4).astype(np.float32), lambda o, k: np.zeros(4))
    assert tot < -85 and 50 < t < 200
    # random torques: falls as well
    r = [_run(sim, rng.rand(4).astype(np.float32), lambda o, k: rng.uniform(-1, 1, 4))[0] for _ in range(5)]
    assert np.mean(r) < -80
    # joints hold: hip and knee anchors coincide to Box2D's tolerances while the walker stands / falls -- except right
    # after a time-of-impact sub-step, which moves ONE body without its joints (Box2D does the same) and leaves the error
    # to the next steps' position solver.  The first of those is the reset itself: gym creates the legs 0.53 m off their
    # hip anchors, the first world step snaps them down, the feet are stopped at the ground, and the knees are 0.23 m
    # apart until the hull has been lifted (3 steps; gym's walker makes the same little hop at the start)
    sim.reset(rng.rand(4).astype(np.float32))
    lh = 34 / 30
    gaps = []
    for k in range(60):
        sim.step(np.array([0.3, -0.3, -0.3, 0.3]))
        b, _, info = sim.debug()
        worst = 0.0
        for up, lo in ((1, 2), (3, 4)):
            pu = b[up][:2] + np.array([np.sin(b[up][2]) * lh / 2, -np.cos(b[up][2]) * lh / 2])     # upper leg's lower end
            pl = b[lo][:2] + np.array([-np.sin(b[lo][2]) * lh / 2, np.cos(b[lo][2]) * lh / 2])     # lower leg's upper end
            worst = max(worst, float(np.hypot(*(pu - pl))))
        gaps.append(worst)
    gaps = np.array(gaps)
    assert gaps[0] > 0.1 and gaps[:3].max() < 0.3 and gaps[3:].max() < 0.1, gaps
    assert np.mean(gaps[3:] < 0.03) > 0.9, gaps
    # the feet start ON the pad, not in it (continuous collision): lowest leg vertex within the contact slop of the ground
    sim.reset(rng.rand(4).astype(np.float32))
    b, terrain, _ = sim.debug()
    feet = [b[lo][1] - np.cos(b[lo][2]) * lh / 2 - abs(np.sin(b[lo][2])) * 0.8 * 8 / 30 / 2 for lo in (2, 4)]
    assert all(abs(f - terrain[10]) < 0.03 for f in feet), (feet, terrain[10])


def test_legs_never_sink_into_the_terrain():
    """A property no twin-source comparison gives: over 1500 steps of random torques (walkers stumbling, falling, being
    stopped by time-of-impact sub-steps) no corner of a leg is ever below the terrain polyline -- the contact skin
    (2 x b2_polygonRadius minus the slop) keeps the core shapes apart.  Also: every joint that is beyond a limit by more than
    0.1 rad is back within 0.1 rad of it at most three steps later (the limits are soft; the one large excursion is the
    snap of the first step after reset)."""
    rng = np.random.RandomState(2)
    LH, LW, STEP = 34 / 30, 8 / 30, 14 / 30
    lims = [(-0.8, 1.1), (-1.6, -0.1), (-0.8, 1.1), (-1.6, -0.1)]
    worst, steps, streak, longest = 0.0, 0, 0, 0
    for ep in range(10):
        sim = co.WalkerSim()
        sim.reset(rng.rand(4).astype(np.float32))
        streak = 0
        for t in range(300):
            a = rng.uniform(-1, 1, 4) if ep % 2 else np.tanh(rng.randn(4))
            _, _, done = sim.step(a)
            b, terr, _ = sim.debug()
            steps += 1
            for k in (1, 2, 3, 4):
                cx, cy, an = (float(v) for v in b[k][:3])
                w = (LW if k in (1, 3) else 0.8 * LW) / 2
                for sx in (-1, 1):
                    for sy in (-1, 1):
                        px = cx + np.cos(an) * sx * w - np.sin(an) * sy * LH / 2
                        py = cy + np.sin(an) * sx * w + np.cos(an) * sy * LH / 2
                        i = int(np.floor(px / STEP))
                        if 0 <= i < 199:
                            f = (px - i * STEP) / STEP
                            worst = max(worst, float(terr[i] * (1 - f) + terr[i + 1] * f) - py)
            ang = [b[1][2] - b[0][2], b[2][2] - b[1][2], b[3][2] - b[0][2], b[4][2] - b[3][2]]
            beyond = max(max(lo - x, x - hi) for (lo, hi), x in zip(lims, ang)) > 0.1
            streak = streak + 1 if beyond else 0
            longest = max(longest, streak)
            if done:
                break
    assert steps > 800
    assert worst <= 1e-4, worst                                     # observed: 0.0 (never below the line)
    assert longest <= 6, longest                                    # observed: 3 consecutive steps (the snap after reset)


def test_population_rollout_entry_point():
    rng = np.random.RandomState(5)
    P = co.param_count(24, 4, False)
    assert P == 932
    theta = (rng.randn(6, P) * 0.5).astype(np.float32)
    init = rng.rand(6, 2, 4).astype(np.float32)
    fit, ep, steps = co.rollout_walker(theta, init, 2, 120)
    assert fit.shape == (6,) and ep.shape == (6, 2) and steps.max() <= 120 and steps.min() >= 1
    fit2, _, _ = co.rollout_walker(theta, init, 2, 120)
    assert np.array_equal(fit, fit2)


# ---- the float32 world against an independently written float64 integration (oracle/walker64.c) ---------------------------

_LH, _LD, _HULL_LC = 34 / 30, -8 / 30, np.array([-0.02003643, -0.00564663])


def _joint_gaps(b):
    """largest distance between the two anchors of a joint, from bodies[5][centre of mass x, y, angle, ...]"""
    def pt(k, lx, ly, lc=(0.0, 0.0)):
        c, s = np.cos(b[k][2]), np.sin(b[k][2])
        x, y = lx - lc[0], ly - lc[1]
        return b[k][:2] + np.array([c * x - s * y, s * x + c * y])
    g = []
    for up, lo in ((1, 2), (3, 4)):
        g.append(np.hypot(*(pt(0, 0, _LD, _HULL_LC) - pt(up, 0, _LH / 2))))
        g.append(np.hypot(*(pt(up, 0, -_LH / 2) - pt(lo, 0, _LH / 2))))
    return max(g)


def test_float32_walker_tables_equal_an_independent_derivation():
    """Mass, inertia, centre of mass and mixed friction of the five bodies: ses_b2_shapes.h (generated, Box2D's float32
    operation order) against the shoelace formulas of oracle/walker64.c over gym's polygons in double precision."""
    from oracle.walker64 import Walker64
    w = Walker64()
    w.reset(np.full(200, 400 / 30 / 4))
    _, _, own = w.debug()                                       # mass, inertia, local centre x, y
    tab = co.walker_body_props().astype(np.float64)
    assert np.allclose(tab[:, 0], own[:, 0], rtol=3e-7) and np.allclose(tab[:, 1], own[:, 1], rtol=3e-7)
    assert np.abs(tab[:, 2:4] - own[:, 2:4]).max() < 2e-8
    assert np.allclose(tab[:, 4], [np.sqrt(0.1 * 2.5)] + [np.sqrt(0.2 * 2.5)] * 4, rtol=1e-7)
    assert abs(own[0, 0] - 5.422222) < 1e-5 and abs(own[1, 0] - 0.302222) < 1e-6 and abs(own[2, 0] - 0.241778) < 1e-6


def test_independent_float64_walker_envelope():
    """The float32 world against oracle/walker64.c -- an integration written independently of it (double precision, own mass
    properties, one generic constraint row type solved to convergence instead of 180 Gauss-Seidel iterations, point
    contacts, no manifolds, no warm starting, no time-of-impact pass) that shares only INPUTS: the terrain, the actions and
    the configuration it adopts.  The CPU build of the float32 world is bit-identical to the device kernels
    (tests/test_gpu_walker.py, test_gpu_envs.py), so this bounds the device trajectories too.

    A walker stands on its feet from the first step and two solvers part company exponentially once a foot slips or lands
    a step apart (measured: the hull angles of two open-loop runs stay within 0.05 rad of each other for a median of 42
    steps), so the envelope is LOCAL: along float32 trajectories -- 30 episodes of piecewise-constant random torques -- the
    float64 integration adopts the float32 state wherever its joints are closed (anchor gaps below 0.5 mm; a time-of-impact
    sub-step moves one body without its joints and Box2D repairs that over the next steps: how, is that solver's path) and
    both advance ONE step.  Windows are sorted by what the step contains.  ENVELOPE (observed in brackets):
      on the adopted state    observation formulas (hull, joints) within 1e-6 [1e-7]; the ten lidar fractions within 2e-5
                              [8e-6]: same pose, two ray casts;
      flight, joints free     226 windows.  Velocity observations within 1e-4 [max 2.5e-5, median 5e-7]: masses, inertias, joint
      or at a limit           rows and motors agree to rounding; hull position / angle within 2e-4 [5.0e-5]; joint angles within
                              3e-3 [9.2e-4: what is left of the anchor gaps is closed along different paths]; reward within 1e-3
                              [2.3e-4];
      feet on the ground,     330 windows.  Velocity observations: median below 2e-3 [8.8e-4], 90 % below 1.5e-2 [5.2e-3; the
      no contact made or      largest 0.40: a foot that sticks in one integration and slides in the other]; joint angles: median
      lost in the step        below 2e-3 [4.2e-4], 90 % below 1.5e-2 [4.6e-3]; hull position / angle: 90 % below 3e-3 [8.0e-4];
                              reward: 90 % below 6e-3 [1.9e-3] -- corner contacts against two-point manifolds, 180 warm-started
                              iterations against convergence;
      any step                the episode ends in the same step in all but 1 % of the 1275 windows [8].
    GLOBAL, where the outcome is not chaotic -- the torque-free collapse of 30 walkers from the adopted start: the hull
    touches down within 8 steps of each other [6, median 2] after 114-126 steps, return within 1.0 [0.29] of -91.7, final
    hull x within 0.2 [0.09].  Open loop with random torques the two runs must still agree on the hull angle to 0.05 rad for
    15 steps at least in every run [22] and for 30 in the median [42]."""
    from oracle.walker64 import Walker64
    rng = np.random.RandomState(1)
    a32, a64 = co.WalkerSim(), Walker64()
    vel_ix, ang_ix = [1, 2, 3, 5, 7, 10, 12], [4, 6, 9, 11]
    rows, ends_apart, windows = [], 0, 0
    for ep in range(30):
        u = rng.rand(4).astype(np.float32)
        a32.reset(u)
        terr = a32.debug()[1].astype(np.float64)
        a64.reset(terr, float(u[0]))
        acts = np.repeat(np.tanh(rng.randn(40, 4)), 8, axis=0)
        t, d32, o32 = 0, False, None
        while t < 300 and not d32:
            b0 = a32.debug()[0].astype(np.float64)
            if t < 4 or _joint_gaps(b0) > 5e-4:
                o32, _, d32 = a32.step(acts[t])
                t += 1
                continue
            o64 = a64.adopt(b0)
            if o32 is not None:
                assert np.abs(o32[:14] - o64[:14])[[0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12]].max() < 1e-6
                assert np.abs(o32[14:] - o64[14:]).max() < 2e-5, (ep, t)
            i0, c0 = a32.debug()[2], a64.debug()[1]["contact"]
            o32, r32, d32 = a32.step(acts[t])
            o64, r64, d64 = a64.step(acts[t])
            t += 1
            windows += 1
            ends_apart += int(d32 != d64)
            if d32 or d64:
                continue
            i1, c1 = a32.debug()[2], a64.debug()[1]["contact"]
            event = (i1["manifolds_per_leg"] != i0["manifolds_per_leg"] or c1 != c0 or i1["limits"] != i0["limits"]
                     or _joint_gaps(a32.debug()[0].astype(np.float64)) > 5e-4)
            contact = sum(i0["manifolds_per_leg"]) > 0 or sum(c0) > 0
            b32, b64 = a32.debug()[0].astype(np.float64), a64.debug()[0]
            rows.append((event, contact, np.abs(b32[0][:3] - b64[0][:3]).max(), np.abs(o32[ang_ix] - o64[ang_ix]).max(),
                         np.abs(o32[vel_ix] - o64[vel_ix]).max(), abs(r32 - r64)))
    rows = np.array(rows, dtype=np.float64)
    assert windows > 1200 and ends_apart <= 0.01 * windows, (windows, ends_apart)
    flight = rows[(rows[:, 0] == 0) & (rows[:, 1] == 0)]
    ground = rows[(rows[:, 0] == 0) & (rows[:, 1] == 1)]
    assert len(flight) > 150 and len(ground) > 250, (len(flight), len(ground))
    assert flight[:, 4].max() < 1e-4 and flight[:, 2].max() < 2e-4 and flight[:, 3].max() < 3e-3 and flight[:, 5].max() < 1e-3, flight.max(0)
    assert np.median(ground[:, 4]) < 2e-3 and np.percentile(ground[:, 4], 90) < 1.5e-2, np.percentile(ground[:, 4], [50, 90])
    assert np.median(ground[:, 3]) < 2e-3 and np.percentile(ground[:, 3], 90) < 1.5e-2, np.percentile(ground[:, 3], [50, 90])
    assert np.percentile(ground[:, 2], 90) < 3e-3 and np.percentile(ground[:, 5], 90) < 6e-3, np.percentile(ground[:, [2, 5]], 90, axis=0)

    def pair(acts, rng_u):
        a32.reset(rng_u)
        a64.reset(a32.debug()[1].astype(np.float64), float(rng_u[0]))
        for t in range(6):
            a32.step(acts[t])
        a64.adopt(a32.debug()[0])
        tot, n, d, together = [0.0, 0.0], [0, 0], [False, False], None
        for t in range(6, 300):
            for k, sim in enumerate((a32, a64)):
                if not d[k]:
                    _, r, d[k] = sim.step(acts[t])
                    tot[k] += r
                    n[k] = t + 1
            if together is None and (any(d) or abs(float(a32.debug()[0][0][2]) - float(a64.debug()[0][0][2])) > 0.05):
                together = t
            if all(d):
                break
        return n, tot, (float(a32.debug()[0][0][0]), float(a64.debug()[0][0][0])), together if together is not None else 300

    rng = np.random.RandomState(3)
    for ep in range(30):                                            # the torque-free collapse
        n, tot, x, _ = pair(np.zeros((320, 4)), rng.rand(4).astype(np.float32))
        assert 105 <= min(n) and max(n) <= 135 and abs(n[0] - n[1]) <= 8, (ep, n)
        assert abs(tot[0] - tot[1]) < 1.0 and -93.5 < min(tot) and max(tot) < -90.0, (ep, tot)
        assert abs(x[0] - x[1]) < 0.2, (ep, x)
    horizon = [pair(np.repeat(np.tanh(rng.randn(40, 4)), 8, axis=0), rng.rand(4).astype(np.float32))[3] for _ in range(30)]
    assert min(horizon) >= 15 and np.median(horizon) >= 30, (min(horizon), np.median(horizon))

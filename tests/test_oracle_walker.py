"""BipedalWalker-v3 on the CPU (oracle/ses_walker_env.h over the Box2D-style world of oracle/ses_b2.h): behavioural
checks -- gym and Box2D are not here to compare with (parity unpinned, see the headers)."""
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

"""OPTIONAL, skipped in this image (neither gym nor Box2D is installed, and there is no network): where
`gym[box2d]` of the reference's era (0.18-0.21: 4-tuple step, `env.np_random` a RandomState) IS installed, the CPU build of
the float32 Box2D-style world is flown next to gym's own LunarLanderContinuous-v2 on the same inputs and held to the
envelope that tests/test_oracle_lander.py holds it to against the independent float64 integration.

gym draws three things from `env.np_random`: the 12 terrain heights, the two components of the initial force, and two
engine-dispersion numbers per step.  The world under test takes the first two from its reset row (ses_lander_env.h:
lander_reset_state) and the third from Philox keyed by the row (oracle/lander64.py: dispersion) -- so a scripted
`np_random` that hands gym exactly those numbers, in the order gym asks for them, puts both on the same episode.
Written without being able to run it: a gym whose API differs from the one described skips instead of failing."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle.lander64 import dispersion

gym = pytest.importorskip("gym")
pytest.importorskip("Box2D")

H_HALF = 400.0 / 30.0 / 2.0                                      # VIEWPORT_H / SCALE / 2


class ScriptedRandom:
    """Stands in for env.np_random: uniform() replays the numbers of one reset row."""

    def __init__(self, u16):
        u = np.asarray(u16, dtype=np.float32)
        self.key = tuple(int(k) for k in u[14:16].view(np.uint32))
        self.heights = (u[2:14].astype(np.float64) * H_HALF)
        self.force = [2000.0 * float(u[0]) - 1000.0, 2000.0 * float(u[1]) - 1000.0]
        self.step, self.half = 0, 0

    def uniform(self, low=0.0, high=1.0, size=None):
        if size is not None:                                     # the terrain: uniform(0, H / 2, size=(CHUNKS + 1,))
            assert tuple(np.atleast_1d(size)) == (12,) and low == 0
            return self.heights.copy()
        if self.force:                                           # the initial force: two scalars in (-1000, 1000)
            return self.force.pop(0)
        d = dispersion(self.key[0], self.key[1], self.step)[self.half]   # then two per step in (-1, 1)
        self.half ^= 1
        self.step += self.half == 0
        return d


def test_float32_world_against_gym_lunar_lander():
    try:
        env = gym.make("LunarLanderContinuous-v2").unwrapped
        if not hasattr(env, "np_random"):
            pytest.skip("this gym has no env.np_random to script")
    except Exception as exc:                                     # pragma: no cover
        pytest.skip(f"gym cannot make LunarLanderContinuous-v2 here: {exc}")
    rng = np.random.RandomState(0)
    sim = co.LanderSim()
    worst = np.zeros(6)
    for ep in range(20):
        u = rng.rand(16).astype(np.float32)
        env.np_random = ScriptedRandom(u)
        try:
            first = env.reset()
        except TypeError as exc:                                 # pragma: no cover
            pytest.skip(f"this gym's reset() does not draw from a scripted np_random: {exc}")
        o_gym = np.asarray(first[0] if isinstance(first, tuple) else first, dtype=np.float64)
        o_sim = sim.reset(u)
        assert np.abs(o_gym[:6] - o_sim[:6]).max() < 2e-3, (ep, o_gym, o_sim)      # both after gym's leg snap
        acts = np.repeat(np.tanh(rng.randn(30, 2) * 1.2), 10, axis=0)
        c_gym = c_sim = None
        for t in range(300):
            a = acts[t].astype(np.float32)
            out = env.step(a)
            o_gym, d_gym = np.asarray(out[0], dtype=np.float64), bool(out[2])
            o_sim, _, d_sim = sim.step(float(a[0]), float(a[1]))
            if c_gym is None and (o_gym[6] or o_gym[7] or d_gym):
                c_gym = t
            if c_sim is None and (o_sim[6] or o_sim[7] or d_sim):
                c_sim = t
            if c_gym is not None or c_sim is not None:
                break
            worst = np.maximum(worst, np.abs(o_gym[:6] - o_sim[:6]))
        assert c_gym is not None and c_sim is not None and abs(c_gym - c_sim) <= 1, (ep, c_gym, c_sim)
    print("max |gym - float32 world| over the flights, per observation component:", worst)
    # the envelope of tests/test_oracle_lander.py::test_independent_float64_lander_envelope (flight part)
    assert (worst < np.array([1.5e-3, 1.5e-3, 5e-3, 5e-3, 3e-3, 1.5e-2])).all(), worst

"""LunarLander-lite: the C oracle against fixture G8 (reference RolloutWorker + reference GRU GymEnvModel with
the continuous tanh head over the build's lander env, POMDP mask) and sanity properties of the reduced physics."""
import json
import os

import numpy as np

from oracle import c_oracle as co
from oracle.lander_env import LunarLanderLiteEnv


def test_g8_returns_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_lander.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g8_lander.json")))
    assert g["theta"].shape[1] == meta["P"] == co.param_count(8, 4, True) == 6756
    fit, ep, steps = co.rollout_lander(g["theta"], g["init"], meta["E"], meta["max_step"])
    # continuous control: actions differ from torch's at the 1e-6 level, returns follow smoothly
    np.testing.assert_allclose(fit.astype(np.float64), g["returns"], rtol=2e-6, atol=1e-4)
    assert steps.max() <= 300 and steps.min() >= 1


def _fly(sim, u, controller, limit=1000):
    obs = sim.reset(u)
    total, t, done = 0.0, 0, False
    while not done and t < limit:
        a0, a1 = controller(obs)
        obs, r, done = sim.step(a0, a1)
        total += r
        t += 1
    return total, t, obs


def test_reduced_physics_behaves_like_a_lander():
    rng = np.random.RandomState(0)
    sim = co.LanderSim()

    def heuristic(obs):                     # the classic PD landing heuristic
        angle_targ = np.clip(obs[0] * 0.5 + obs[2] * 1.0, -0.4, 0.4)
        hover_targ = 0.55 * abs(obs[0])
        angle_todo = (angle_targ - obs[4]) * 0.5 - obs[5] * 1.0
        hover_todo = (hover_targ - obs[1]) * 0.5 - obs[3] * 0.5
        if obs[6] or obs[7]:
            angle_todo, hover_todo = 0.0, -obs[3] * 0.5
        a = np.clip([hover_todo * 20 - 1, -angle_todo * 20], -1, 1)
        return float(a[0]), float(a[1])

    landed = [_fly(sim, rng.rand(16).astype(np.float32), heuristic) for _ in range(10)]
    assert min(r for r, _, _ in landed) > 200                      # soft landings end with +100 (asleep)
    assert all(o[6] == 1 and o[7] == 1 for _, _, o in landed)       # on both legs
    fall = [_fly(sim, rng.rand(16).astype(np.float32), lambda o: (0.0, 0.0)) for _ in range(10)]
    assert max(r for r, _, _ in fall) < 0 and max(t for _, t, _ in fall) < 120   # free fall crashes (-100)
    rand = [_fly(sim, rng.rand(16).astype(np.float32), lambda o: tuple(rng.uniform(-1, 1, 2)), 300)
            for _ in range(10)]
    assert np.mean([r for r, _, _ in rand]) < -50


def test_env_object_protocol_and_pomdp_mask():
    init = np.random.RandomState(1).rand(2, 16).astype(np.float32)
    env = LunarLanderLiteEnv(init, max_step=20, pomdp=True)
    s = env.reset()
    o = s["0"]["state"]
    assert o.shape == (8,) and o[2] == 0 and o[3] == 0 and o[5] == 0 and o[1] > 1.0
    t, done = 0, False
    while not done:
        s, r, done, _ = env.step({"0": np.array([1.0, 0.0, 0.3, -0.3], np.float32)})
        t += 1
    assert t == 20                                                  # truncated by max_step (gym_wrapper.py:37-39)

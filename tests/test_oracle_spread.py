"""simple_spread (multi-agent): the C oracle against fixture G7 -- returns of the REFERENCE RolloutWorker with
one deepcopy of the reference GymEnvModel per agent (utils.py:4-8) over the build's simple_spread env."""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle.spread_env import SimpleSpreadF32Env


@pytest.mark.parametrize("n_agents", [2, 3])
def test_g7_team_returns_match_reference(golden_dir, n_agents):
    g = np.load(os.path.join(golden_dir, "g7_spread.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g7_spread.json")))[f"n{n_agents}"]
    theta, init = g[f"n{n_agents}_theta"], g[f"n{n_agents}_init"]
    assert theta.shape[1] == meta["P"] == co.param_count(6 * n_agents, 5, False)
    fit, ep = co.rollout_spread(theta, init, meta["E"], n_agents, meta["max_cycles"])
    assert np.abs(fit.astype(np.float64) - g[f"n{n_agents}_returns"]).max() <= 1e-4
    assert (ep < 0).all()                                        # rewards are negative distances / collisions


@pytest.mark.parametrize("n_agents", [2, 3])
def test_g7t_trained_team_returns_match_reference(golden_dir, n_agents):
    """G7's teams are random; G7t: 20 elite checkpoints of product runs (conf/simplespread.yaml and the three-agent BASELINE shape)
    and 40 seeded perturbations of them per shape, returns from the reference's RolloutWorker + wrap_agentid: 120 / 120 within
    1e-4 (observed 3.8e-6: no argmax of 5 actions flipped in 37 500 agent-steps)."""
    g = np.load(os.path.join(golden_dir, "g7t_spread_trained.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g7t_spread_trained.json")))[f"n{n_agents}"]
    theta, init = g[f"n{n_agents}_theta"], g[f"n{n_agents}_init"]
    assert theta.shape == (meta["N"], co.param_count(6 * n_agents, 5, False)) and meta["N"] >= 60
    fit, _ = co.rollout_spread(theta, init, meta["E"], n_agents)
    assert np.abs(fit.astype(np.float64) - g[f"n{n_agents}_returns"]).max() <= 1e-4


def test_env_protocol_and_observation_layout():
    init = np.array([[0.5, -0.25, -0.5, 0.75, 0.1, 0.2, -0.3, -0.4]], np.float32)   # 2 agents, 2 landmarks
    env = SimpleSpreadF32Env(init, n_agents=2)
    assert env.get_agent_ids() == ["agent_0", "agent_1"]
    obs = env.reset()
    o0 = obs["agent_0"]["state"]
    assert o0.shape == (12,) and o0.dtype == np.float32
    np.testing.assert_allclose(o0, [0, 0, 0.5, -0.25, 0.1 - 0.5, 0.2 + 0.25, -0.3 - 0.5, -0.4 + 0.25,
                                    -0.5 - 0.5, 0.75 + 0.25, 0, 0], atol=1e-7)
    done, t, total = False, 0, 0.0
    while not done:
        obs, r, done, _ = env.step({"agent_0": np.array(2), "agent_1": np.array(0)})
        total += r
        t += 1
    assert t == 25 and total < 0
    assert obs["agent_0"]["state"][2] > 0.5                      # agent 0 pushed in +x


def test_collision_penalty_and_contact_force():
    # two agents almost on top of each other: local reward -1 each, strong repulsion
    st = np.array([0.0, 0.0, 0.05, 0.0, 0, 0, 0, 0, 1.0, 1.0, -1.0, -1.0], np.float32)
    r = co.spread_step(2, st, np.array([0, 0], np.int32))
    assert st[0] < 0.0 and st[2] > 0.05                          # pushed apart along x
    glob = -(np.hypot(st[0] - 1, st[1] - 1) if np.hypot(st[0] - 1, st[1] - 1) < np.hypot(st[2] - 1, st[3] - 1)
             else np.hypot(st[2] - 1, st[3] - 1))
    assert r < 2 * 0.5 * glob                                    # below the pure-distance part: collision terms applied

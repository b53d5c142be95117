"""Step-wise device envs (ses_env_reset / ses_env_step_generic, csrc/ses_envs.hip) and the wrappers on top of them:
every transition of every env bit-equal to the oracle's env objects, one lane per env (64 different worlds per wave);
and the reference's playback loop (test.py:53-63) running against GymWrapper / PettingzooWrapper for every supported
env, episode returns equal to the same loop over the oracle's env objects."""
import os
from copy import deepcopy

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle.cartpole_env import CartPoleF32Env
from oracle.lander_env import LunarLanderEnv
from oracle.spread_env import SimpleSpreadF32Env

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("pomdp", [False, True])
def test_lander_transitions_bit_equal_to_the_oracle_env(pomdp):
    from ses import HipES
    n, T = 96, 140                                                   # two waves, one of them half full
    es = HipES("LunarLanderContinuous-v2", 8, 4, False, False, pomdp=pomdp, max_step=300, eval_ep_num=1)
    init = es.init_states_uniform(5, 1, 0, n)[:, 0].contiguous()
    state, obs = es.env_reset(init)
    assert state.shape == (n, es.env_state_bytes()) and obs.shape == (n, 8)
    sims = [co.LanderSim() for _ in range(n)]
    mask = np.array([0, 0, 1, 1, 0, 1, 0, 0], bool) if pomdp else np.zeros(8, bool)
    want = np.stack([s.reset(u) for s, u in zip(sims, init.cpu().numpy())])
    want[:, mask] = 0
    assert np.array_equal(bits(obs.cpu().numpy()), bits(want))
    rng = np.random.RandomState(3)
    alive = np.ones(n, bool)
    crashed = landed_contact = 0
    for t in range(T):
        act = np.tanh(rng.randn(n, 4) * 1.5).astype(np.float32)
        act[: n // 3, 0] = np.abs(act[: n // 3, 0])                   # a third of them keeps the main engine on: long flights
        o, r, d = es.env_step_generic(state, dev(act))
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        for i in np.flatnonzero(alive):
            wo, wr, wd = sims[i].step(float(act[i, 0]), float(act[i, 1]))
            wo[mask] = 0
            assert np.array_equal(bits(o[i]), bits(wo)), (t, i)
            assert bits(r[i:i + 1])[0] == bits(np.float32(wr))[0] and bool(d[i]) == wd, (t, i, r[i], wr)
            landed_contact += int(wo[6] + wo[7] > 0) if not pomdp else 0
            if wd:
                alive[i] = False
                crashed += 1
    assert crashed > n // 2 and alive.sum() > 0                       # crashes and survivors in one population of lanes
    es.close()


def test_walker_transitions_bit_equal_to_the_oracle_env():
    from ses import HipES
    n, T = 40, 70
    es = HipES("BipedalWalker-v3", 24, 4, False, False, max_step=300, eval_ep_num=1)
    init = es.init_states_uniform(2, 0, 0, n)[:, 0].contiguous()
    state, obs = es.env_reset(init)
    sims = [co.WalkerSim() for _ in range(n)]
    want = np.stack([s.reset(u) for s, u in zip(sims, init.cpu().numpy())])
    assert np.array_equal(bits(obs.cpu().numpy()), bits(want))
    rng = np.random.RandomState(1)
    alive = np.ones(n, bool)
    for t in range(T):
        act = np.tanh(rng.randn(n, 4)).astype(np.float32)
        o, r, d = es.env_step_generic(state, dev(act))
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        for i in np.flatnonzero(alive):
            wo, wr, wd = sims[i].step(act[i])
            assert np.array_equal(bits(o[i]), bits(wo)), (t, i)
            assert bits(r[i:i + 1])[0] == bits(np.float32(wr))[0] and bool(d[i]) == wd, (t, i)
            alive[i] = alive[i] and not wd
    assert 0 < alive.sum() < n
    es.close()


@pytest.mark.parametrize("na", [2, 3])
def test_spread_transitions_bit_equal_to_the_oracle_env(na):
    from ses import HipES
    n = 70
    es = HipES("simple_spread", 6 * na, 5, True, False, max_step=25, eval_ep_num=1, n_agents=na)
    init = es.init_states_uniform(9, 4, 0, n)[:, 0].contiguous()
    state, obs = es.env_reset(init)
    assert obs.shape == (n, na * 6 * na)
    envs = [SimpleSpreadF32Env(u[None], n_agents=na) for u in init.cpu().numpy()]
    first = [e.reset() for e in envs]
    want = np.stack([np.concatenate([f[a]["state"] for a in e.agents]) for f, e in zip(first, envs)])
    assert np.array_equal(bits(obs.cpu().numpy()), bits(want))
    rng = np.random.RandomState(na)
    for t in range(25):
        act = rng.randint(0, 5, size=(n, na)).astype(np.int32)
        o, r, d = es.env_step_generic(state, dev(act))
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        for i, e in enumerate(envs):
            tr, wr, wd, _ = e.step({a: np.array(act[i, j]) for j, a in enumerate(e.agents)})
            wo = np.concatenate([tr[a]["state"] for a in e.agents])
            assert np.array_equal(bits(o[i]), bits(wo)), (t, i)
            assert bits(r[i:i + 1])[0] == bits(np.float32(wr))[0] and bool(d[i]) == wd == (t == 24), (t, i)
    es.close()


def test_cartpole_transitions_and_pomdp_mask():
    from ses import HipES
    n = 130
    for pomdp in (False, True):
        es = HipES("CartPole-v1", 4, 2, True, False, pomdp=pomdp, max_step=500, eval_ep_num=1)
        init = es.init_states_uniform(1, 0, 0, n)[:, 0].contiguous()
        state, obs = es.env_reset(init)
        st = [init[:, k].cpu().numpy().copy() for k in range(4)]
        ret, status = np.zeros(n, np.float32), np.zeros(n, np.uint32)
        rng = np.random.RandomState(0)
        for t in range(60):
            act = rng.randint(0, 2, size=n).astype(np.int32)
            o, r, d = es.env_step_generic(state, dev(act))
            co.cartpole_step_soa(1, 0, st[0], st[1], st[2], st[3], act, ret, status)      # fixed-length: state always advances
            want = np.stack(st, axis=1).astype(np.float32)
            if pomdp:
                want[:, [1, 3]] = 0
            assert np.array_equal(bits(o.cpu().numpy()), bits(want)), t
            assert (r.cpu().numpy() == 1.0).all()
            term = (np.abs(st[0]) > 2.4) | (np.abs(st[2]) > 0.20943951)
            assert np.array_equal(d.cpu().numpy().astype(bool), term), t
        es.close()


def playback(env, network, episodes):
    """The reference's test.py loop (test.py:45-63) without the renderer."""
    agent_ids = env.get_agent_ids()
    out = []
    for _ in range(episodes):
        models = {}
        for agent_id in agent_ids:
            models[agent_id] = deepcopy(network)
            models[agent_id].eval()
            models[agent_id].reset()
        obs = env.reset()
        done, episode_reward, ep_step = False, 0, 0
        while not done:
            actions = {}
            for k, model in models.items():
                s = obs[k]["state"][np.newaxis, ...]
                actions[k] = model(s)
            obs, r, done, _ = env.step(actions)
            episode_reward += r
            ep_step += 1
        out.append((episode_reward, ep_step))
    return out


@pytest.mark.parametrize("name", ["cartpole_pomdp_gru", "lunarlander_openai", "bipedalwalker", "simplespread"])
def test_the_reference_playback_loop_runs_on_the_wrappers(name):
    """builder.build_env + builder.build_network of a shipped config, a random policy, the reference's playback loop over
    the WRAPPER (device env, one transition per launch) and over the oracle's env object fed the same reset rows:
    identical episode lengths and returns."""
    import yaml
    import builder
    from ses import HipES
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.load(open(os.path.join(ROOT, "simple-es_amd", "conf", name + ".yaml")), Loader=yaml.FullLoader)
    env = builder.build_env(cfg["env"])
    net = builder.build_network(cfg["network"])
    rng = np.random.RandomState(7)
    net.load_flat((rng.randn(net.param_count()) * 0.4).astype(np.float32))
    episodes = 2 if name == "bipedalwalker" else 3
    got = playback(env, net, episodes)
    # the oracle env objects replay the rows the wrapper drew: (seed_env = 0, episode k)
    probe = env._device()
    rows = np.stack([probe.init_states_uniform(0, k, 0, 1)[0, 0].cpu().numpy() for k in range(episodes)])
    if name == "cartpole_pomdp_gru":
        ref_env = CartPoleF32Env(rows, max_step=cfg["env"]["max_step"], pomdp=True)
    elif name == "lunarlander_openai":
        ref_env = LunarLanderEnv(rows, max_step=cfg["env"]["max_step"], pomdp=cfg["env"]["pomdp"])
    elif name == "simplespread":
        ref_env = SimpleSpreadF32Env(rows, n_agents=env.n_agents, max_step=cfg["env"]["max_step"])
    else:
        ref_env = None
    assert all(steps >= 1 for _, steps in got)
    if ref_env is not None:
        want = playback(ref_env, net, episodes)
        assert [s for _, s in got] == [s for _, s in want], (got, want)
        assert np.array_equal(np.array([r for r, _ in got], np.float64).view(np.uint64),
                              np.array([r for r, _ in want], np.float64).view(np.uint64)), (got, want)
    else:                                                            # BipedalWalker: against the fused rollout kernel instead
        es = HipES("BipedalWalker-v3", 24, 4, False, False, max_step=env.horizon, eval_ep_num=1)
        for k, (ret, steps) in enumerate(got):
            init = es.init_states_uniform(0, k, 0, 1)
            _, ep_ret, ep_steps = es.rollout(dev(net.flat()[None, :]), init, want_episodes=True)
            assert int(ep_steps[0, 0]) == steps and float(ep_ret[0, 0]) == ret, (k, got, ep_ret, ep_steps)
        es.close()
    env.close() if hasattr(env, "close") else None

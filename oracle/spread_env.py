"""CPU simple_spread env object speaking the reference's PettingzooWrapper protocol
(envs/pettingzoo_wrapper.py:22-61): reset() -> {agent: {"state": obs}}, step({agent: action}) ->
(dict, team_reward, done, {}), get_agent_ids().  TEST INFRASTRUCTURE (oracle).

Physics: oracle/ses_oracle.c::spread_step (fp32; pettingzoo is absent -> parity unpinned).
Initial states are an explicit input (rows of [agent positions 2n | landmark positions 2n]) replayed
round-robin, because the reference never seeds its envs.
"""
import numpy as np

from . import c_oracle


class SimpleSpreadF32Env:
    name = "simple_spread"

    def __init__(self, init_states, n_agents=2, max_step="None", max_cycles=25):
        self.n = n_agents
        self.init_states = np.asarray(init_states, dtype=np.float32).reshape(-1, 4 * n_agents)
        self.max_step = max_step
        self.max_cycles = max_cycles
        self.agents = [f"agent_{i}" for i in range(n_agents)]
        self.curr_step = 0
        self._next = 0
        self._st = None

    def rewind(self, index=0):
        self._next = index

    def get_agent_ids(self):
        return list(self.agents)

    def _obs_dict(self):
        return {a: {"state": c_oracle.spread_obs(self.n, self._st, i)} for i, a in enumerate(self.agents)}

    def reset(self):
        self.curr_step = 0
        s0 = self.init_states[self._next % len(self.init_states)]
        self._next += 1
        n = self.n
        self._st = np.zeros(6 * n, dtype=np.float32)
        self._st[: 2 * n] = s0[: 2 * n]
        self._st[4 * n:] = s0[2 * n:]
        return self._obs_dict()

    def step(self, action):
        self.curr_step += 1
        acts = np.array([int(np.asarray(action[a])) for a in self.agents], dtype=np.int32)
        total_r = c_oracle.spread_step(self.n, self._st, acts)
        done = self.curr_step >= self.max_cycles                 # pettingzoo max_cycles: all agents done together
        if self.max_step != "None":
            if self.curr_step >= self.max_step or done:
                done = True
        out = self._obs_dict()
        for a in self.agents:
            out[a].update(reward=total_r / self.n, done=done, info={})
        return out, total_r, done, {}

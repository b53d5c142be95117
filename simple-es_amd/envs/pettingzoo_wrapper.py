"""PettingzooWrapper -- the reference's multi-agent env adapter (envs/pettingzoo_wrapper.py:6-64) backed by
the device simple_spread kernels.

The reference hard-codes `simple_spread_v2.env(N=2)` (pettingzoo_wrapper.py:9); `n_agents` keeps that
default and also allows 3 (BASELINE.json configs[4]).  waterworld / multiwalker need polygon-polygon contacts /
a 242-wide observation and are not built: they raise instead of falling back to a CPU env.

As with GymWrapper, the population rollout never steps this object: ESLoop hands the whole shard to the
fused kernel.  reset() / step() keep the reference's dict protocol for single-team use (the reference's test.py
loop): one-lane launches of ses_env_reset / ses_env_step_generic, the same spread_obs / spread_step device functions
the fused rollout calls.
"""
import numpy as np
import torch

MAX_CYCLES = 25   # pettingzoo mpe default max_cycles: every agent is done after 25 cycles


class PettingzooWrapper:
    def __init__(self, name, max_step=None, n_agents=2):
        if name != "simple_spread":
            raise NotImplementedError(f"env {name!r} has no gfx950 kernel in this build (available: simple_spread); "
                                      "there is no CPU/pettingzoo fallback")
        if n_agents not in (2, 3):
            raise NotImplementedError("simple_spread kernels are instantiated for 2 or 3 agents")
        self.name = name
        self.max_step = max_step
        self.n_agents = n_agents
        self.pomdp = False
        self.horizon = MAX_CYCLES if max_step in (None, "None") else min(int(max_step), MAX_CYCLES)
        self.agents = [f"agent_{i}" for i in range(n_agents)]
        self.curr_step = 0
        self.seed_env = 0
        self._episode = 0
        self._dev = None
        self._state = None

    def get_agent_ids(self):
        return list(self.agents)

    def _device(self):
        if self._dev is None:
            from ses import HipES
            self._dev = HipES(self.name, 6 * self.n_agents, 5, True, False, max_step=self.horizon, eval_ep_num=1,
                              n_agents=self.n_agents)
        return self._dev

    def _transitions(self, obs):
        per = obs.view(self.n_agents, 6 * self.n_agents).cpu().numpy()
        return {agent: {"state": per[i].copy()} for i, agent in enumerate(self.agents)}

    def reset(self):
        """pettingzoo_wrapper.py:22-31: {agent: {"state": obs}}; positions from row (seed_env, episode) of the ENV_INIT stream."""
        dev = self._device()
        self.curr_step = 0
        init = dev.init_states_uniform(self.seed_env, self._episode, 0, 1)[:, 0].contiguous()
        self._episode += 1
        self._state, obs = dev.env_reset(init)
        return self._transitions(obs[0])

    def step(self, action):
        """pettingzoo_wrapper.py:33-58: every agent's action is set, the world advances one cycle; per-agent transitions,
        the TEAM reward (sum over the agents) and done = all agents done or curr_step >= max_step."""
        dev = self._device()
        self.curr_step += 1
        acts = torch.tensor([[int(np.asarray(action[a])) for a in self.agents]], dtype=torch.int32, device=dev.device)
        obs, reward, done = dev.env_step_generic(self._state, acts)
        total_r, d = float(reward[0].item()), bool(int(done[0].item()))
        out = self._transitions(obs[0])
        for tr in out.values():
            # per-agent rewards of a cycle are equal shares of local + global terms in the fused kernel's accounting; the
            # protocol's consumers (loop.py:123, test.py:61) use the team total returned below
            tr.update(reward=total_r / self.n_agents, done=d, info={})
        if self.max_step != "None" and self.max_step is not None:
            if self.curr_step >= int(self.max_step) or d:
                d = True
        return out, total_r, d, {}

    def render(self):
        raise NotImplementedError("no renderer on the device path")

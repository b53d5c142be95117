"""PettingzooWrapper -- the reference's multi-agent env adapter (envs/pettingzoo_wrapper.py:6-64) backed by
the device simple_spread kernels.

The reference hard-codes `simple_spread_v2.env(N=2)` (pettingzoo_wrapper.py:9); `n_agents` keeps that
default and also allows 3 (BASELINE.json configs[4]).  waterworld / multiwalker need a Box2D-class rigid
body solver and are not built: they raise instead of falling back to a CPU env.

As with GymWrapper, the population rollout never steps this object: ESLoop hands the whole shard to the
fused kernel.  reset()/step() keep the reference's dict protocol for single-team use and run the same
device functions through a population-of-one... they are not implemented step-wise on the device yet, so
they raise NotImplementedError (playback uses RolloutWorker, which is implemented).
"""

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

    def get_agent_ids(self):
        return list(self.agents)

    def reset(self):
        raise NotImplementedError("step-wise simple_spread is not exposed; use ESLoop / RolloutWorker (fused device rollout)")

    def step(self, action):
        raise NotImplementedError("step-wise simple_spread is not exposed; use ESLoop / RolloutWorker (fused device rollout)")

    def render(self):
        raise NotImplementedError("no renderer on the device path")

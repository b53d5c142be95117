"""CPU LunarLanderContinuous-v2 env object speaking the reference's GymWrapper protocol (envs/gym_wrapper.py:23-48)
with the LunarLanderPOMDP mask (obs[2] = obs[3] = obs[5] = 0, gym_wrapper.py:57-66).  TEST INFRASTRUCTURE.

Physics: oracle/ses_lander_env.h over the Box2D-style world of oracle/ses_b2.h (three bodies, two revolute joints,
polygon / terrain-edge contacts, 180 + 60 solver iterations per step).  gym and Box2D are absent from the reference tree
and from this image: parity with them is unpinned, see the headers.  Initial states: rows of 16 uniforms replayed
round-robin."""
import numpy as np

from . import c_oracle


class LunarLanderEnv:
    name = "LunarLanderContinuous-v2"

    def __init__(self, init_states, max_step=300, pomdp=True):
        self.init_states = np.asarray(init_states, dtype=np.float32).reshape(-1, 16)
        self.max_step = max_step
        self.pomdp = pomdp
        self.curr_step = 0
        self._next = 0
        self._sim = c_oracle.LanderSim()

    def rewind(self, index=0):
        self._next = index

    def get_agent_ids(self):
        return ["0"]

    def _wrap(self, obs):
        if self.pomdp:
            obs = obs.copy()
            obs[2] = 0
            obs[3] = 0
            obs[5] = 0
        return {"0": {"state": obs}}

    def reset(self):
        self.curr_step = 0
        u = self.init_states[self._next % len(self.init_states)]
        self._next += 1
        return self._wrap(self._sim.reset(u))

    def step(self, action):
        self.curr_step += 1
        a = np.asarray(action["0"], dtype=np.float32).reshape(-1)     # 4 tanh outputs, the env uses [0] and [1]
        obs, r, d = self._sim.step(float(a[0]), float(a[1]))
        if self.max_step != "None":
            if self.curr_step >= self.max_step or d:
                d = True
        tr = self._wrap(obs)
        tr["0"].update(reward=r, done=d, info={})
        return tr, r, d, {}

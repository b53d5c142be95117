"""CPU BipedalWalker-v3 env object speaking the reference's GymWrapper protocol (envs/gym_wrapper.py:23-48).  TEST INFRASTRUCTURE.

Physics: simple-es_amd/csrc/ses_walker_env.h over the Box2D-style world of ses_b2.h, compiled for the host by oracle/Makefile
(five bodies, four revolute joints with limits and motors, polygon / terrain-edge contacts, lidar, 180 + 60 solver iterations per
step, time-of-impact sub-stepping).  gym and Box2D are absent from the reference tree and from this image: parity with them is
unpinned, see the headers.  Initial states: rows of 4 uniforms (initial force, terrain key) replayed round-robin."""
import numpy as np

from . import c_oracle


class BipedalWalkerEnv:
    name = "BipedalWalker-v3"

    def __init__(self, init_states, max_step=300):
        self.init_states = np.asarray(init_states, dtype=np.float32).reshape(-1, 4)
        self.max_step = max_step
        self.curr_step = 0
        self._next = 0
        self._sim = c_oracle.WalkerSim()

    def rewind(self, index=0):
        self._next = index

    def get_agent_ids(self):
        return ["0"]

    def reset(self):
        self.curr_step = 0
        u = self.init_states[self._next % len(self.init_states)]
        self._next += 1
        return {"0": {"state": self._sim.reset(u)}}

    def step(self, action):
        self.curr_step += 1
        a = np.asarray(action["0"], dtype=np.float32).reshape(-1)     # four joint torques in [-1, 1] (tanh head)
        obs, r, d = self._sim.step(a)
        if self.max_step != "None":
            if self.curr_step >= self.max_step or d:
                d = True
        tr = {"0": {"state": obs, "reward": r, "done": d, "info": {}}}
        return tr, r, d, {}

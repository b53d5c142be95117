"""CPU CartPole env objects speaking the reference's env-wrapper protocol.

TEST INFRASTRUCTURE (oracle).  These objects can be handed to the reference's
own `RolloutWorker` / `ESLoop` (learning_strategies/evolution/loop.py:108-125)
in place of `envs.gym_wrapper.GymWrapper` (gym is not installed in this image):

    reset()  -> {"0": {"state": ndarray}}                 gym_wrapper.py:23-30
    step({"0": action}) -> (dict, reward, done, info)     gym_wrapper.py:32-45
    get_agent_ids() -> ["0"]                              gym_wrapper.py:47-48
    truncation rule  done = curr_step >= max_step or d    gym_wrapper.py:37-39
    POMDP mask       obs[1] = obs[3] = 0                  gym_wrapper.py:69-77

The reference never seeds its env, so initial states are an
explicit input here: `reset()` replays rows of `init_states` round-robin.

Two physics variants:
  * CartPoleF32Env   -- the build's definition: fp32 state, the deterministic
                        sincos of oracle/ses_oracle_math.h (calls the C oracle so
                        the arithmetic is the same code the HIP kernel is checked
                        against).
  * CartPoleGym64Env -- gym-faithful float64 restatement (math.sin/cos), used
                        only to quantify how far fp32 physics moves returns.
gym's cartpole.py is third-party and absent here: parity at this boundary is
unpinned; constants follow SURVEY Appendix A.1.
"""
import math

import numpy as np

from . import c_oracle


class _ReplayEnvBase:
    name = "CartPole-v1"

    def __init__(self, init_states, max_step=500, pomdp=False):
        self.init_states = np.asarray(init_states, dtype=np.float32).reshape(-1, 4)
        self.max_step = max_step
        self.pomdp = pomdp
        self.curr_step = 0
        self._next = 0

    def rewind(self, index=0):
        self._next = index

    def get_agent_ids(self):
        return ["0"]

    def _wrap(self, obs):
        if self.pomdp:
            obs = obs.copy()
            obs[1] = 0
            obs[3] = 0
        return {"0": {"state": obs}}

    def reset(self):
        self.curr_step = 0
        s0 = self.init_states[self._next % len(self.init_states)]
        self._next += 1
        self._set_state(s0)
        return self._wrap(self._obs())

    def step(self, action):
        self.curr_step += 1
        a = int(np.asarray(action["0"]))
        d = self._physics(a)
        if self.max_step != "None":
            if self.curr_step >= self.max_step or d:
                d = True
        tr = self._wrap(self._obs())
        tr["0"]["reward"] = 1.0
        tr["0"]["done"] = d
        tr["0"]["info"] = {}
        return tr, 1.0, d, {}

    def close(self):
        pass


class CartPoleF32Env(_ReplayEnvBase):
    def _set_state(self, s0):
        self._x = np.array([s0[0]], dtype=np.float32)
        self._xd = np.array([s0[1]], dtype=np.float32)
        self._th = np.array([s0[2]], dtype=np.float32)
        self._thd = np.array([s0[3]], dtype=np.float32)
        self._ret = np.zeros(1, dtype=np.float32)
        self._status = np.zeros(1, dtype=np.uint32)
        self._act = np.zeros(1, dtype=np.int32)

    def _obs(self):
        return np.array([self._x[0], self._xd[0], self._th[0], self._thd[0]], dtype=np.float32)

    def _physics(self, a):
        self._act[0] = a
        # max_step=0: truncation is applied by step() above, exactly like GymWrapper does
        c_oracle.cartpole_step_soa(c_oracle.MODE_EPISODIC, 0, self._x, self._xd, self._th, self._thd,
                                   self._act, self._ret, self._status)
        return bool(self._status[0] >> 31)


class CartPoleGym64Env(_ReplayEnvBase):
    gravity = 9.8
    masscart = 1.0
    masspole = 0.1
    total_mass = masspole + masscart
    length = 0.5
    polemass_length = masspole * length
    force_mag = 10.0
    tau = 0.02
    theta_threshold_radians = 12 * 2 * math.pi / 360
    x_threshold = 2.4

    def _set_state(self, s0):
        self._s = [float(v) for v in s0]

    def _obs(self):
        return np.array(self._s, dtype=np.float32)

    def _physics(self, a):
        x, x_dot, theta, theta_dot = self._s
        force = self.force_mag if a == 1 else -self.force_mag
        costheta = math.cos(theta)
        sintheta = math.sin(theta)
        temp = (force + self.polemass_length * theta_dot ** 2 * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / (
            self.length * (4.0 / 3.0 - self.masspole * costheta ** 2 / self.total_mass))
        xacc = temp - self.polemass_length * thetaacc * costheta / self.total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * xacc
        theta = theta + self.tau * theta_dot
        theta_dot = theta_dot + self.tau * thetaacc
        self._s = [x, x_dot, theta, theta_dot]
        return bool(x < -self.x_threshold or x > self.x_threshold
                    or theta < -self.theta_threshold_radians or theta > self.theta_threshold_radians)

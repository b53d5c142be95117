"""GymWrapper -- the reference's env adapter (envs/gym_wrapper.py:7-54) backed by the device env kernels.

The reference wraps a gym env object; gym's physics is third-party Python.  Here the env IS a gfx950
kernel: the population rollout never calls this object step by step (ESLoop hands the whole shard to the
fused rollout kernel), it only reads `name / max_step / pomdp`.  `reset()` / `step()` keep the reference's
dict protocol for single-env use (checkpoint playback, the reference's test.py loop) for EVERY supported env:
one-lane launches of ses_env_reset / ses_env_step_generic, i.e. the same device functions the fused rollouts call.
"""
import numpy as np
import torch

SUPPORTED = {"CartPole-v1": dict(num_state=4, num_action=2, discrete=True, time_limit=500),
             "CartPole-v0": dict(num_state=4, num_action=2, discrete=True, time_limit=200),
             # gym's env restated on a Box2D-style world (csrc/ses_lander.h, ses_b2.h) -- gym's own TimeLimit is 1000 steps
             "LunarLanderContinuous-v2": dict(num_state=8, num_action=4, discrete=False, time_limit=1000),
             # gym's env restated on the same world (csrc/ses_walker.h) -- gym's TimeLimit is 1600 steps
             "BipedalWalker-v3": dict(num_state=24, num_action=4, discrete=False, time_limit=1600)}


class GymWrapper:
    def __init__(self, name, max_step=None, pomdp=False, physics="float32"):
        if name not in SUPPORTED:
            raise NotImplementedError(
                f"env {name!r} has no gfx950 kernel in this build (available: {sorted(SUPPORTED)}); "
                "there is no CPU/gym fallback")
        if pomdp and "CartPole" not in name and "LunarLander" not in name:
            raise AssertionError(f"{name} doesn't support POMDP.")
        self.name = name
        self.pomdp = bool(pomdp)
        if physics not in ("float32", "float64"):
            raise ValueError("env.physics must be 'float32' (default) or 'float64' (gym-order CartPole dynamics)")
        self.physics64 = physics == "float64"
        self.spec = SUPPORTED[name]
        # a gym name whose third-party physics is restated here without a pin says so at run time (ESLoop prints it,
        # metrics.jsonl records it): returns are the build's own, not gym + Box2D's bit for bit
        self.variant = "box2d-restated" if ("LunarLander" in name or "BipedalWalker" in name) else None
        # YAML `max_step: None` is the STRING "None" in the reference (gym_wrapper.py:37); accept both.
        limit = self.spec["time_limit"]
        self.max_step = max_step
        self.horizon = limit if max_step in (None, "None") else min(int(max_step), limit)
        self.curr_step = 0
        self.seed_env = 0
        self._episode = 0
        self._dev = None
        self._state = None

    def get_agent_ids(self):
        return ["0"]

    # ---- single-env protocol (playback) ---------------------------------------------------------
    def _device(self):
        if self._dev is None:
            from ses import HipES
            sp = self.spec
            # (env.physics: float64 exists in the fused rollouts only: the library refuses a step-wise reset of such a handle)
            self._dev = HipES(self.name, sp["num_state"], sp["num_action"], sp["discrete"], False, pomdp=self.pomdp,
                              max_step=self.horizon, eval_ep_num=1, physics64=self.physics64)
        return self._dev

    def reset(self):
        """gym_wrapper.py:23-30: {"0": {"state": obs}}; the reset distribution is this library's (the reference never seeds
        its env): episode k of this object draws row (seed_env, k) of the ENV_INIT Philox stream."""
        dev = self._device()
        self.curr_step = 0
        init = dev.init_states_uniform(self.seed_env, self._episode, 0, 1)[:, 0].contiguous()     # [1, init_dim]
        self._episode += 1
        self._state, obs = dev.env_reset(init)
        return {"0": {"state": obs[0].cpu().numpy()}}

    def step(self, action):
        """gym_wrapper.py:32-45: ({"0": transition}, r, done, info); done = env done or curr_step >= max_step."""
        dev = self._device()
        self.curr_step += 1
        a = np.asarray(action["0"])
        if self.spec["discrete"]:
            act = torch.tensor([int(a)], dtype=torch.int32, device=dev.device)
        else:
            act = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).reshape(1, -1)).to(dev.device)
        obs, reward, done = dev.env_step_generic(self._state, act)
        s, r, d = obs[0].cpu().numpy(), float(reward[0].item()), bool(int(done[0].item()))
        if self.max_step != "None" and self.max_step is not None:
            if self.curr_step >= int(self.max_step) or d:
                d = True
        if self.curr_step >= self.spec["time_limit"]:          # gym's own TimeLimit wrapper around the registered env
            d = True
        tr = {"state": s, "reward": r, "done": d, "info": {}}
        return {"0": tr}, r, d, {}

    def render(self):
        raise NotImplementedError("no renderer on the device path")

    def close(self):
        if self._dev is not None:
            self._dev.close()
            self._dev = None

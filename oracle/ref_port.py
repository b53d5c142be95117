"""Structural CPU port of the reference's rollout path -- the `cpu_baseline` leg of bench.py.

TEST / MEASUREMENT INFRASTRUCTURE (oracle).  Same shape of computation as the reference:
one `multiprocessing.Pool(process_num)` per generation, an order-preserving `map` over offspring
(learning_strategies/evolution/loop.py:66-79), and per offspring a batch-1 torch CPU forward per env
step (networks/neural_network.py:20-36) against a Python env object (loop.py:108-125).  The reference
files themselves never travel to the GPU box, so this restatement is what gets timed there; its
per-offspring returns are pinned to fixture G5 (tests/test_ref_port.py).
"""
import multiprocessing as mp
import os
import time

import numpy as np
import torch
from torch import nn

from .cartpole_env import CartPoleF32Env


class PolicyNet(nn.Module):
    """fc1(S->32) tanh [GRU(32,32) tanh] fc2(32->A); argmax or tanh head."""

    def __init__(self, num_state, num_action, discrete_action, gru):
        super().__init__()
        self.discrete_action = discrete_action
        self.fc1 = nn.Linear(num_state, 32)
        self.gru = nn.GRU(32, 32) if gru else None
        self.fc2 = nn.Linear(32, num_action)
        self.h = None
        self.reset()

    def reset(self):
        self.h = torch.zeros(1, 1, 32) if self.gru is not None else None

    def load_flat(self, vec):
        off = 0
        with torch.no_grad():
            for p in self.parameters():
                n = p.numel()
                p.copy_(torch.from_numpy(np.asarray(vec[off:off + n], dtype=np.float32)).view_as(p))
                off += n

    @torch.no_grad()
    def act(self, obs):
        x = torch.tanh(self.fc1(torch.from_numpy(obs).float().view(1, 1, -1)))
        if self.gru is not None:
            x, self.h = self.gru(x, self.h)
            x = torch.tanh(x)
        y = self.fc2(x).view(-1)
        if self.discrete_action:
            return torch.argmax(torch.softmax(y, dim=0)).numpy()
        return torch.tanh(y).numpy()


class FixedLengthCartPole(CartPoleF32Env):
    """Synthetic-benchmark variant: termination masked, an episode always lasts max_step steps
    (data-independent work)."""

    def step(self, action):
        tr, r, d, info = super().step(action)
        d = self.curr_step >= self.max_step
        return tr, r, d, info


def rollout_worker(task):
    env, net_cfg, vec, episodes = task
    torch.set_num_threads(1)
    net = PolicyNet(*net_cfg)
    net.load_flat(vec)
    total, steps = 0.0, 0
    for _ in range(episodes):
        states = env.reset()
        net.reset()
        done = False
        while not done:
            action = net.act(states["0"]["state"][np.newaxis, ...])
            states, r, done, _ = env.step({"0": action})
            total += r
            steps += 1
    return total / episodes, steps


def run_generation(theta, init_states, episodes, max_step, process_num, fixed_length=False, net_cfg=(4, 2, True, False),
                   pomdp=False):
    """One generation's rollout phase.  Returns (returns[N], env_steps, seconds)."""
    cls = FixedLengthCartPole if fixed_length else CartPoleF32Env
    env = cls(init_states, max_step=max_step, pomdp=pomdp)
    tasks = [(env, net_cfg, theta[i], episodes) for i in range(theta.shape[0])]
    from . import c_oracle
    c_oracle.lib()                                       # built and loaded BEFORE the fork: no worker builds anything
    t0 = time.perf_counter()
    if process_num > 1:
        pool = mp.Pool(process_num)
        try:
            # a worker that dies takes its task with it and map() would wait for ever: bounded instead
            out = pool.map_async(rollout_worker, tasks).get(timeout=600)
            pool.close()
        except BaseException:
            pool.terminate()
            raise
        finally:
            pool.join()
    else:
        out = [rollout_worker(t) for t in tasks]
    dt = time.perf_counter() - t0
    return np.array([o[0] for o in out]), int(sum(o[1] for o in out)), dt


def time_baseline(process_num=None, offspring_per_proc=6, episodes=5, max_step=500, seed=0):
    """Bounded sample of the benchmark workload (CartPole, MLP, theta = 0.1*eps, fixed-length episodes)."""
    t_begin = time.perf_counter()
    process_num = process_num or os.cpu_count() or 1
    n = max(16, offspring_per_proc * process_num)
    rng = np.random.RandomState(seed)
    theta = (rng.standard_normal((n, 226)) * 0.1).astype(np.float32)
    init = np.random.RandomState(0).uniform(-0.05, 0.05, (episodes, 4)).astype(np.float32)
    _, steps, dt = run_generation(theta, init, episodes, max_step, process_num, fixed_length=True)
    # the same worker in-process on ONE core (loop.py:76, process_num == 1), smaller sample
    _, steps1, dt1 = run_generation(theta[:8], init, episodes, max_step, 1, fixed_length=True)
    # for scale: the compiled C oracle (same arithmetic as the kernels) on one core
    from . import c_oracle
    t0 = time.perf_counter()
    c_oracle.rollout_cartpole(theta[:64], init, episodes, max_step, mode=c_oracle.MODE_FIXED_LENGTH)
    c_rate = 64 * episodes * max_step / (time.perf_counter() - t0)
    model, physical, sockets = "unknown", None, None
    try:
        with open("/proc/cpuinfo") as f:
            text = f.read()
        model = next(line.split(":", 1)[1].strip() for line in text.splitlines() if line.startswith("model name"))
        # physical cores = distinct (socket, core id) pairs; logical CPUs = processor entries (SMT siblings share a pair)
        pairs, sock, core = set(), None, None
        for line in text.splitlines() + [""]:
            if line.startswith("physical id"):
                sock = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if sock is not None and core is not None:
                    pairs.add((sock, core))
                sock = core = None
        if pairs:
            physical, sockets = len(pairs), len({p[0] for p in pairs})
    except (OSError, StopIteration):
        pass
    return {"value": steps / dt, "unit": "env-steps/s", "cores": process_num, "kind": "port",
            "cores_logical": os.cpu_count(), "cores_physical": physical, "sockets": sockets,
            "cores_note": "`cores` = worker processes of the pool = logical CPUs the host shows (SMT siblings count twice); "
                          "cores_physical = distinct (socket, core id) pairs in /proc/cpuinfo",
            "seconds": time.perf_counter() - t_begin,
            "sample": f"{n} offspring x {episodes} episodes x {max_step} fixed-length steps = {steps} env-steps "
                      f"in {dt:.2f}s, mp.Pool({process_num}), batch-1 torch forward + Python CartPole per step",
            "value_1_process": steps1 / dt1, "c_oracle_1_core": c_rate, "cpu_model": model}

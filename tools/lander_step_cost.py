#!/usr/bin/env python3
"""Cost of ONE LunarLander world step on the device, by lane density (development helper): the step-wise entry
(ses_env_step_generic, one lane = one env, 64 different worlds per wave) timed at n = 64 ... 131072 envs, in free flight
(main engine on) and after the craft have come down.  Tells apart what a step costs a wave (latency) from what the chip
sustains (throughput) -- the two numbers the C3 rollout's time is made of."""
import json, os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES

es = HipES("LunarLanderContinuous-v2", 8, 4, False, False, pomdp=True, max_step=300, eval_ep_num=1)
for n in (64, 1024, 16384, 65536, 131072):
    init = es.init_states_uniform(3, 0, 0, n)[:, 0].contiguous()
    state, obs = es.env_reset(init)
    up = torch.zeros(n, 4, device="cuda"); up[:, 0] = 0.3          # gentle main engine: slow descent / hover
    off = torch.zeros(n, 4, device="cuda"); off[:, 0] = -1.0
    row = {"n_envs": n, "waves": (n + 63) // 64}
    for label, act, steps in (("flight_us_per_step", up, 40), ("ground_phase_us_per_step", off, 160)):
        for _ in range(5):
            es.env_step_generic(state, act)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            o, r, d = es.env_step_generic(state, act)
        e1.record(); e1.synchronize()
        row[label] = round(e0.elapsed_time(e1) * 1e3 / steps, 1)
        row[label.replace("us_per_step", "legs_down_frac")] = round(float(((o[:, 6] + o[:, 7]) > 0).float().mean()), 3)
    row["flight_env_steps_per_s"] = round(n / (row["flight_us_per_step"] * 1e-6))
    print(json.dumps(row), flush=True)

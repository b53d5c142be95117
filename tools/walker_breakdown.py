#!/usr/bin/env python3
"""Where the BipedalWalker rollout's time goes: rollout time by horizon (the same 4096 x 5 population cut at 50 ... 300
steps), next to the number of envs still alive at each horizon.  usage: walker_breakdown.py [offspring] [lpe ...]"""
import json, os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lpes = [int(a) for a in sys.argv[2:]] or [0]
for lpe in lpes:
    row = {"lanes_per_env": lpe, "offspring": n}
    for h in (25, 50, 100, 150, 200, 250, 300):
        es = HipES("BipedalWalker-v3", 24, 4, False, False, max_step=h, eval_ep_num=5)
        es.set_tuning("box2d_lanes_per_env", lpe)
        theta = es.perturb(es.zeros(es.P), 2.0, 0, 0, 0, n)
        init = es.init_states_uniform(0, 0, 0, n)
        fit = es.empty(n)
        es.rollout(theta, init, fitness=fit); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); es.rollout(theta, init, fitness=fit); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        _, _, steps = es.rollout(theta, init, want_episodes=True)
        row[f"ms_{h}"] = round(statistics.median(ts), 2)
        row[f"alive_at_{h}"] = int((steps >= h).sum().item())
        es.close()
    print(json.dumps(row), flush=True)

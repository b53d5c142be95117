#!/usr/bin/env python3
"""Rollout time of config C3 (LunarLanderContinuous-v2 POMDP, GRU, 4096 offspring x 5 episodes x <= 300 steps) and of the
MLP lander config; one JSON line each (development helper; tools/bench_configs.py is the recorded form)."""
import json, os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES

def run(name, gru, pomdp, n, sigma, reps=3):
    es = HipES("LunarLanderContinuous-v2", 8, 4, False, gru, pomdp=pomdp, max_step=300, eval_ep_num=5)
    mu = es.zeros(es.P)
    theta = es.perturb(mu, sigma, 0, 0, 0, n)
    init = es.init_states_uniform(0, 0, 0, n)
    fit = es.empty(n)
    es.rollout(theta, init, fitness=fit); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); es.rollout(theta, init, fitness=fit); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    _, _, steps = es.rollout(theta, init, want_episodes=True)
    total = int(steps.sum().item())
    ms = statistics.median(ts)
    print(json.dumps({"config": name, "offspring": n, "rollout_ms": ms, "env_steps": total, "mean_episode_steps": total / (n * 5),
                      "env_steps_per_s": total / (ms * 1e-3), "fitness_mean": float(fit.mean())}), flush=True)
    es.close()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
which = sys.argv[2] if len(sys.argv) > 2 else "both"
if which in ("both", "gru"):
    run("C3 LunarLanderContinuous-v2 POMDP GRU", True, True, n, 0.168)
if which in ("both", "mlp"):
    run("LunarLanderContinuous-v2 MLP", False, False, n, 2.0)

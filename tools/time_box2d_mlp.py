#!/usr/bin/env python3
"""Rollout time of the Box2D MLP kernels (LunarLanderContinuous-v2, BipedalWalker-v3; 5 episodes x <= 300 steps) per
lanes-per-env setting; one JSON line each.  usage: time_box2d_mlp.py [offspring] [lpe ...]   (lpe 0 = library's choice)"""
import json, os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES

def run(env, S, n, sigma, lpe, reps=3):
    es = HipES(env, S, 4, False, False, max_step=300, eval_ep_num=5)
    es.set_tuning("box2d_lanes_per_env", lpe)
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
    print(json.dumps({"env": env, "offspring": n, "lanes_per_env": lpe, "rollout_ms": round(ms, 3), "env_steps": total,
                      "mean_episode_steps": round(total / (n * 5), 1), "env_steps_per_s": total / (ms * 1e-3),
                      "fitness_mean": float(fit.mean())}), flush=True)
    es.close()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lpes = [int(a) for a in sys.argv[2:]] or [0]
for lpe in lpes:
    run("LunarLanderContinuous-v2", 8, n, 2.0, lpe)
for lpe in lpes:
    run("BipedalWalker-v3", 24, n, 2.0, lpe)

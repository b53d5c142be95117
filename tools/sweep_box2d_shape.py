#!/usr/bin/env python3
"""Rollout time of the Box2D MLP kernels by wave shape (lanes per env, different envs per wave); one JSON line each.
usage: sweep_box2d_shape.py <lander|walker> <offspring> lpe:epw [lpe:epw ...]      (0:0 = the library's choice)"""
import json, os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES

which, n = sys.argv[1], int(sys.argv[2])
env, S = ("LunarLanderContinuous-v2", 8) if which == "lander" else ("BipedalWalker-v3", 24)
ref = None
for shape in sys.argv[3:]:
    lpe, epw = (int(x) for x in shape.split(":"))
    es = HipES(env, S, 4, False, False, max_step=300, eval_ep_num=5)
    es.set_tuning("box2d_lanes_per_env", lpe)
    es.set_tuning("box2d_envs_per_wave", epw)
    theta = es.perturb(es.zeros(es.P), 2.0, 0, 0, 0, n)
    init = es.init_states_uniform(0, 0, 0, n)
    fit = es.empty(n)
    es.rollout(theta, init, fitness=fit); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); es.rollout(theta, init, fitness=fit); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    same = None
    if ref is None: ref = fit.clone()
    else: same = bool(torch.equal(ref.view(torch.int32), fit.view(torch.int32)))
    print(json.dumps({"env": env, "offspring": n, "lanes_per_env": lpe, "envs_per_wave": epw,
                      "rollout_ms": round(statistics.median(ts), 2), "bits_equal_to_first": same}), flush=True)
    es.close()

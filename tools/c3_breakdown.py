#!/usr/bin/env python3
"""Where the C3 rollout's time goes (development helper): LunarLanderContinuous-v2 POMDP GRU, 4096 offspring x 5
episodes, first-generation policies.  Rollout time by horizon (max_step) and offspring per wave, and the histogram of
episode lengths -- a wave lives as long as its longest episode, the kernel as long as its slowest wave."""
import json, os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def timed(es, theta, init, fit, reps=3):
    es.rollout(theta, init, fitness=fit)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); es.rollout(theta, init, fitness=fit); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


for T in (50, 100, 150, 200, 300):
    es = HipES("LunarLanderContinuous-v2", 8, 4, False, True, pomdp=True, max_step=T, eval_ep_num=5)
    theta = es.perturb(es.zeros(es.P), 0.168, 0, 0, 0, n)
    init = es.init_states_uniform(0, 0, 0, n)
    fit = es.empty(n)
    row = {"max_step": T}
    for g in (1, 2, 4):
        es.set_tuning("lander_offspring_per_wave", g)
        row[f"ms_g{g}"] = round(timed(es, theta, init, fit), 3)
    _, _, steps = es.rollout(theta, init, want_episodes=True)
    s = steps.flatten().cpu()
    row["env_steps"] = int(s.sum())
    row["mean_len"] = round(float(s.float().mean()), 1)
    if T == 300:
        edges = [0, 60, 80, 100, 120, 150, 200, 250, 299, 300]
        row["len_hist"] = {f"<={e}": int((s <= e).sum()) for e in edges[1:]}
        per_off = steps.max(dim=1).values.cpu()
        row["offspring_longest_mean"] = round(float(per_off.float().mean()), 1)
        row["offspring_with_300"] = int((per_off >= 300).sum())
        pair = steps.view(-1, 2, 5).amax(dim=(1, 2)).cpu()
        row["pair_of_offspring_longest_mean"] = round(float(pair.float().mean()), 1)
        row["pairs_with_300"] = int((pair >= 300).sum())
    print(json.dumps(row), flush=True)
    es.close()

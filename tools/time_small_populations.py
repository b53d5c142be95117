#!/usr/bin/env python3
"""The fused CartPole MLP rollout at the per-GPU populations of the strong-scaling line (4096 offspring in total over 1 / 2 / 4 / 8 / 16
GPUs = 4096 ... 256 per GPU), 5 episodes x 500 fixed-length steps, by lanes per env (0 = the library's choice) and, for 8 / 16 lanes,
with the packed step of lone waves forced off (`s`) and on (`p`): us per ses_rollout (rollout kernel + episode-mean kernel)."""
import os, sys, json, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "simple-es_amd")]
from ses import HipES, MODE_FIXED_LENGTH


def time_us(es, n):
    theta = es.perturb(es.zeros(es.P), 0.1, 0, 0, 0, n)
    init = es.init_states_uniform(0, 0, 0, 1, shared=True)[0].contiguous()
    fit = es.empty(n)
    for _ in range(30): es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    return round(statistics.median(ts), 1)


for n in (256, 512, 1024, 2048, 4096):
    row = {"offspring": n}
    for lpe in (0, 4, 8, 16, 32):
        for packed in ((None,) if lpe in (0, 4, 32) else (0, 1)):
            es = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=5, lanes_per_env=lpe)
            if packed is not None:
                es.set_tuning("rollout_packed", packed)
            row[f"lpe{lpe}" + ("" if packed is None else "sp"[packed]) + "_us"] = time_us(es, n)
            es.close()
    print(json.dumps(row), flush=True)

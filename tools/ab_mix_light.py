#!/usr/bin/env python3
"""A/B of the fused CartPole MLP rollout's mixed split in ONE process, interleaved rounds (development helper):
light waves at 8 lanes per env (round 2) against 16 (round 3), plus the pure splits, at the benchmark population and
at the sizes around it; and the small-population case (lanes per env 8 against 16)."""
import json, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES, MODE_FIXED_LENGTH


def timed(es, theta, init, fit, k=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / k


for n in (4096, 96, 800, 1200, 1640, 2048, 2560, 3072, 3584, 5120, 6144, 8192, 9830):
    variants = {}
    if n * 5 > 8192:
        for light in (8, 16):
            es = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=5)
            es.set_tuning("rollout_mix_light", light)
            variants[f"mix light={light}"] = es
    for lpe in (4, 8, 16):
        variants[f"pure lpe={lpe}"] = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=5, lanes_per_env=lpe)
    variants["auto"] = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=5)
    any_es = next(iter(variants.values()))
    theta = any_es.perturb(any_es.zeros(any_es.P), 0.1, 0, 0, 0, n)
    init = any_es.init_states_uniform(0, 0, 0, 1, shared=True)[0]
    fit = any_es.empty(n)
    ref = None
    for name, es in variants.items():
        got = es.rollout(theta, init, mode=MODE_FIXED_LENGTH).cpu()
        ref = got if ref is None else ref
        assert torch.equal(got, ref), name
    for _ in range(30):
        for es in variants.values():
            es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit)
    torch.cuda.synchronize()
    times = {k: [] for k in variants}
    for r in range(5):
        for name, es in variants.items():
            times[name].append(timed(es, theta, init, fit))
    print(json.dumps({"offspring": n, "us": {k: round(statistics.median(v), 1) for k, v in times.items()},
                      "min_us": {k: round(min(v), 1) for k, v in times.items()}}), flush=True)
    for es in variants.values():
        es.close()

#!/usr/bin/env python3
"""Time the replicated fitness loop (rank_center + es_update_philox) at the global populations of 1/2/4/8/16 GPUs."""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES  # noqa: E402


def timed(fn, reps=15):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1))
    return statistics.median(out)


es = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=5)
mu, m, v = es.zeros(es.P), es.zeros(es.P), es.zeros(es.P)
for n in (4096, 8192, 16384, 32768, 65536):
    fit = torch.rand(n, device=es.device) * 500
    w = es.rank_center(fit)[1]
    print(json.dumps({"n": n, "rank_center_us": 1e3 * timed(lambda: es.rank_center(fit)),
                      "es_update_philox_us": 1e3 * timed(lambda: es.es_update_philox(w, 0, 1, 0.05, 0.1, 0.05, mu, m, v,
                                                                                      skip_row0=False))}), flush=True)

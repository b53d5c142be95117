#!/usr/bin/env python3
"""Time of ses_openai_generation (the openai_es fitness loop + next population) on ONE GPU at the global population sizes
a rank sees at 1 / 2 / 4 / 8 / 16 GPUs of weak scaling (4096 rows per rank are perturbed, all n are ranked and enter the
gradient): what every rank adds to a generation besides the collective.  One JSON line per size."""
import json, os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES

es = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=5)
if len(sys.argv) > 1:
    es.set_tuning("es_final_max_chunks", int(sys.argv[1]))      # 0: the update always in its own launch
per = 4096
for world in (1, 2, 4, 8, 16):
    n = per * world
    fit = torch.rand(n, device=es.device)
    a = [es.zeros(es.P) for _ in range(3)]
    b = [es.zeros(es.P) for _ in range(3)]
    theta = es.empty(per, es.P)
    def call(g):
        es.openai_generation(fit, 1, g, 0.05, 0.1, 0.05, a, b, 0.1, g + 1, 0, per, theta_next=theta)
    for g in range(5):
        call(g)
    torch.cuda.synchronize()
    ts = []
    for rep in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for g in range(20):
            call(g)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(json.dumps({"ranks": world, "n_global": n, "rows_perturbed": per, "openai_generation_us": round(statistics.median(ts), 2)}), flush=True)
es.close()

#!/usr/bin/env python3
"""POMDP CartPole GRU rollout (4096 offspring x E episodes x 500 fixed-length steps) by kernel: the VALU lockstep form, the 4x4x1 MFMA
form (ses_gru_mfma4.h, knob gru_mfma4_min_e) and, from 12 episodes, the 16x16x4 MFMA form.  ms per ses_rollout."""
import os, sys, json, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "simple-es_amd")]
from ses import HipES, MODE_FIXED_LENGTH

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for E in (4, 5, 6, 7, 8):
    row = {"offspring": n, "eval_ep_num": E}
    for name, knobs in (("valu_lockstep_ms", {"gru_mfma4_min_e": 0}), ("mfma_4x4x1_ms", {"gru_mfma4_min_e": 1})):
        es = HipES("CartPole-v1", 4, 2, True, True, pomdp=True, max_step=500, eval_ep_num=E)
        es.set_tuning("gru_ep_parallel_max", 0)
        for k, v in knobs.items():
            es.set_tuning(k, v)
        theta = es.perturb(es.zeros(es.P), 0.1, 0, 0, 0, n)
        init = es.init_states_uniform(0, 0, 0, 1, shared=True)[0].contiguous()
        fit = es.empty(n)
        for _ in range(3): es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit)
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 3)
        row[name] = round(statistics.median(ts), 3)
        es.close()
    row["mfma_over_valu"] = round(row["mfma_4x4x1_ms"] / row["valu_lockstep_ms"], 3)
    print(json.dumps(row), flush=True)

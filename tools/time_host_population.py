#!/usr/bin/env python3
"""What a caller pays who keeps the population in HOST memory (the reference does: numpy parameter lists, pickled to the workers):
the headline rollout -- 4096 offspring x 5 episodes x 500 fixed-length steps -- with theta copied from pinned host memory before
every rollout and the fitness vector copied back after it, against the device-resident path the product runs (bench.py's `value`
is the device-resident rate by contract; this is the PCIe-inclusive one for DESIGN 6).  One JSON line."""
import json, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES, MODE_FIXED_LENGTH

n, E, T = 4096, 5, 500
es = HipES("CartPole-v1", 4, 2, True, False, max_step=T, eval_ep_num=E)
theta = es.perturb(es.zeros(es.P), 0.1, 0, 0, 0, n)
init = es.init_states_uniform(0, 0, 0, 1, shared=True)[0]
fit = es.empty(n)
host_theta = theta.cpu().pin_memory()
host_fit = torch.empty(n, dtype=torch.float32).pin_memory()


def timed(fn, reps=9, iters=50):
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / iters)
    return statistics.median(out)


def resident():
    es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit)


def through_pcie():
    theta.copy_(host_theta, non_blocking=True)
    es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit)
    host_fit.copy_(fit, non_blocking=True)


def copy_only():
    theta.copy_(host_theta, non_blocking=True)


with torch.cuda.stream(es.stream):
    for f in (resident, through_pcie, copy_only):
        f()
    torch.cuda.synchronize()
    r, p, c = timed(resident), timed(through_pcie), timed(copy_only)
steps = n * E * T
print(json.dumps({"workload": "4096 x 5 x 500 fixed-length CartPole rollout", "theta_bytes": host_theta.numel() * 4,
                  "rollout_resident_ms": round(r, 4), "rollout_with_h2d_theta_and_d2h_fitness_ms": round(p, 4),
                  "h2d_theta_only_ms": round(c, 4), "h2d_gb_per_s": round(host_theta.numel() * 4 / (c * 1e-3) / 1e9, 1),
                  "env_steps_per_s_resident": steps / (r * 1e-3), "env_steps_per_s_pcie_inclusive": steps / (p * 1e-3)}))
es.close()

#!/usr/bin/env python3
"""C3 (LunarLanderContinuous-v2 POMDP, GRU, 4096 offspring x 5 episodes, first-generation policies): what could ANY
re-bucketing of offspring into waves buy?  ("waves re-bucketed ONCE at step ~50 by any-env-in-contact,
so that flight-only waves stop paying the union of contact rows".)

A wave's step costs the union of its lanes' paths.  A re-bucketing at step ~50 can only use what is known at step 50; an
upper bound on its gain is a grouping that KNOWS each env's future.  That bound needs no new kernel: offspring are
independent, so the rows of (theta, init) are permuted on the host, the same rollout kernel forms its waves from consecutive
rows, and the returns -- permuted back -- must equal the natural order's bit for bit.  Groupings:

    natural          rows as drawn
    by_onset         sorted by the step at which the offspring's FIRST env touches the ground (leg contact or crash),
                     then by its last: waves whose envs come down together, flight-only waves stay flight-only longest
    by_longest       sorted by the offspring's longest episode: the 300-step survivors share waves
    by_onset_blocks  by_onset inside blocks of 1024 rows only (what a re-bucketing within a workgroup-sized window could do)

Contact onsets come from a step-wise replay (ses_policy_forward + ses_env_step_generic: leg-contact flags of the
observation, done flag).  Rollout time by horizon, 4 and 2 offspring per wave, median of 5 launches, groupings interleaved."""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES  # noqa: E402

n, E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 5


def onsets(theta, init):
    """[n, E] step index (1-based) of the first ground contact of every env, 301 = never, from a step-wise replay"""
    sw = HipES("LunarLanderContinuous-v2", 8, 4, False, True, pomdp=True, max_step=300, eval_ep_num=1)
    rows = theta.repeat_interleave(E, dim=0).contiguous()                  # one policy row per env
    state, obs = sw.env_reset(init.reshape(n * E, 16).contiguous())
    hid = torch.zeros(n * E, 32, device="cuda")
    first = torch.full((n * E,), 301, dtype=torch.int32, device="cuda")
    alive = torch.ones(n * E, dtype=torch.bool, device="cuda")
    for t in range(1, 301):
        o = obs.clone()
        o[:, [2, 3, 5]] = 0.0                                               # LunarLanderPOMDP (gym_wrapper.py:61-66)
        _, _, act = sw.policy_forward(rows, o, hid)
        obs, _, done = sw.env_step_generic(state, act)
        touch = ((obs[:, 6] + obs[:, 7]) > 0) | (done != 0)
        first = torch.where(alive & touch & (first == 301), torch.full_like(first, t), first)
        alive = alive & (done == 0)
        if not bool(alive.any()):
            break
    sw.close()
    return first.view(n, E)


def main():
    es = HipES("LunarLanderContinuous-v2", 8, 4, False, True, pomdp=True, max_step=300, eval_ep_num=E)
    theta = es.perturb(es.zeros(es.P), 0.168, 0, 0, 0, n)
    init = es.init_states_uniform(0, 0, 0, n)
    fit0, _, steps = es.rollout(theta, init, want_episodes=True)
    on = onsets(theta, init)
    key_on = on.min(dim=1).values.long() * 1000 + on.max(dim=1).values.long().clamp(max=999)
    orders = {"natural": torch.arange(n, device="cuda"),
              "by_onset": torch.argsort(key_on, stable=True),
              "by_longest": torch.argsort(steps.max(dim=1).values.long(), stable=True)}
    blocks = [b * 1024 + torch.argsort(key_on[b * 1024:(b + 1) * 1024], stable=True) for b in range((n + 1023) // 1024)]
    orders["by_onset_blocks"] = torch.cat(blocks)
    print(json.dumps({"n": n, "first_contact_step": {"median": float(on.float().median()), "p10": float(on.float().quantile(0.1)),
                                                     "p90": float(on.float().quantile(0.9))},
                      "offspring_first_contact": {"median": float(on.min(dim=1).values.float().median()),
                                                  "all_envs_down_median": float(on.max(dim=1).values.float().median())}}), flush=True)
    perm = {k: (theta[o].contiguous(), init[o].contiguous(), o) for k, o in orders.items()}
    for T in (50, 100, 150, 200, 300):
        h = HipES("LunarLanderContinuous-v2", 8, 4, False, True, pomdp=True, max_step=T, eval_ep_num=E)
        row = {"max_step": T}
        ref = None
        for g in (4, 2):
            h.set_tuning("lander_offspring_per_wave", g)
            ts = {k: [] for k in perm}
            fit = h.empty(n)
            for rep in range(6):
                for k, (th, ini, o) in perm.items():                       # interleaved: drift of the box is common to all
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    h.rollout(th, ini, fitness=fit)
                    e1.record()
                    e1.synchronize()
                    if rep:
                        ts[k].append(e0.elapsed_time(e1))
                    else:                                                  # first pass: the bits
                        back = torch.empty_like(fit)
                        back[o] = fit
                        if ref is None:
                            ref = back.clone()
                        assert torch.equal(back.view(torch.int32), ref.view(torch.int32)), (T, g, k)
            for k in perm:
                row[f"{k}_g{g}_ms"] = round(statistics.median(ts[k]), 3)
        if T == 300:
            assert torch.equal(ref.view(torch.int32), fit0.view(torch.int32))
        print(json.dumps(row), flush=True)
        h.close()
    print("returns bit-equal to the natural order for every grouping, horizon and wave shape")


if __name__ == "__main__":
    main()

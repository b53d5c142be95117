#!/usr/bin/env python3
"""Secondary measurements: one rollout (and the replicated fitness loop) at the shape of every BASELINE.json config.

bench.py stays the headline (configs[1]); this prints one JSON line per configuration with the kernel time of
`ses_rollout` (HIP events on the handle's stream, median of --reps) and the env-steps actually taken.  Populations are
theta = sigma * Philox noise around a zero parent, resets are per offspring, like the first generation of a run.
    python tools/bench_configs.py > gpurun_out/configs.jsonl
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from ses import HipES, MODE_EPISODIC, MODE_FIXED_LENGTH  # noqa: E402


def timed(fn, reps):
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


def rollout_case(name, es, n, sigma, mode, reps, steps_per_episode=None):
    mu = es.zeros(es.P)
    theta = es.perturb(mu, sigma, 0, 0, 0, n)
    init = es.init_states_uniform(0, 0, 0, n)
    fit = es.empty(n)
    ms = timed(lambda: es.rollout(theta, init, mode=mode, fitness=fit), reps)
    if steps_per_episode is None:
        _, _, ep_steps = es.rollout(theta, init, mode=mode, want_episodes=True)
        env_steps = int(ep_steps.sum().item()) if mode == MODE_EPISODIC else n * es.E * es.max_step
    else:
        env_steps = n * es.E * steps_per_episode
    f = fit.float()
    print(json.dumps({"config": name, "offspring": n, "P": es.P, "eval_ep_num": es.E, "sigma": sigma,
                      "mode": "episodic" if mode == MODE_EPISODIC else "fixed_length", "rollout_ms": ms,
                      "env_steps": env_steps, "env_steps_per_s": env_steps / (ms * 1e-3),
                      "fitness_mean": float(f.mean()), "fitness_max": float(f.max())}), flush=True)
    return theta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--offspring", type=int, default=4096)
    args = ap.parse_args()
    n, reps = args.offspring, args.reps

    es = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=5)
    rollout_case("C2 CartPole-v1 MLP (bench workload)", es, n, 0.1, MODE_FIXED_LENGTH, reps)
    rollout_case("C2 CartPole-v1 MLP, episodic, first-generation policies", es, n, 0.1, MODE_EPISODIC, reps)
    rollout_case("C4 shard: CartPole-v1 MLP, 8192 offspring per GPU", es, 2 * n, 0.1, MODE_FIXED_LENGTH, reps)
    # the replicated fitness loop of C4 at the global population
    big = 16 * n
    fit = torch.rand(big, device=es.device)
    mu, m, v = es.zeros(es.P), es.zeros(es.P), es.zeros(es.P)
    w = es.rank_center(fit)[1]
    ms_rank = timed(lambda: es.rank_center(fit), reps)
    ms_upd = timed(lambda: es.es_update_philox(w, 0, 1, 0.05, 0.1, 0.05, mu, m, v, skip_row0=False), reps)
    print(json.dumps({"config": "C4 fitness loop at the global population (every rank)", "offspring": big,
                      "rank_center_ms": ms_rank, "es_update_philox_ms": ms_upd}), flush=True)
    es.close()

    es = HipES("CartPole-v1", 4, 2, True, True, pomdp=True, max_step=500, eval_ep_num=5)
    rollout_case("POMDP CartPole-v1 GRU (README learning result)", es, n, 0.1, MODE_FIXED_LENGTH, reps)
    es.close()

    es = HipES("LunarLanderContinuous-v2", 8, 4, False, True, pomdp=True, max_step=300, eval_ep_num=5)
    rollout_case("C3 LunarLanderContinuous-v2 POMDP GRU (conf/lunarlander_openai.yaml)", es, n, 0.168, MODE_EPISODIC, reps)
    es.close()
    es = HipES("LunarLanderContinuous-v2", 8, 4, False, False, pomdp=False, max_step=300, eval_ep_num=5)
    rollout_case("LunarLanderContinuous-v2 MLP (conf/lunarlander.yaml)", es, n, 2.0, MODE_EPISODIC, reps)
    es.close()

    es = HipES("BipedalWalker-v3", 24, 4, False, False, max_step=300, eval_ep_num=5)
    rollout_case("BipedalWalker-v3 MLP (conf/bipedalwalker.yaml)", es, n, 2.0, MODE_EPISODIC, min(reps, 3))
    es.close()

    for agents, S in ((3, 18), (2, 12)):
        es = HipES("simple_spread", S, 5, True, False, max_step=25, eval_ep_num=5, n_agents=agents)
        rollout_case(f"C5 simple_spread, {agents} agents (world steps; {agents} forwards each)", es, n, 1.0, MODE_EPISODIC,
                     reps, steps_per_episode=25)
        es.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Randomised parity sweep: many small random rollout configurations, HIP path vs the C oracle, bit for bit.
Not part of the test suite (minutes of oracle time); run on a GPU box:  python tools/fuzz_parity.py --cases 300"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
from oracle import c_oracle as co  # noqa: E402
from ses import HipES  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    tally = {}
    for case in range(args.cases):
        kind = rng.choice(["mlp", "mlp", "gru", "gru_mfma", "lander", "lander_mlp", "walker", "spread"])
        mode = int(rng.randint(0, 2))
        shared = bool(rng.randint(0, 2))
        sigma = float(rng.choice([0.05, 0.3, 1.0, 3.0]))
        if kind == "mlp":
            n, E, T = int(rng.choice([1, 3, 17, 64, 129, 700, 1700, 2100])), int(rng.randint(1, 8)), int(rng.choice([1, 7, 60, 200]))
            pomdp, lpe, p64 = bool(rng.randint(0, 2)), int(rng.choice([0, 0, 1, 2, 4, 8])), bool(rng.rand() < 0.15)
            es = HipES("CartPole-v1", 4, 2, True, False, pomdp=pomdp, max_step=T, eval_ep_num=E, lanes_per_env=lpe, physics64=p64)
            theta = (rng.randn(n, 226) * sigma).astype(np.float32)
            init = rng.uniform(-0.05, 0.05, (E, 4) if shared else (n, E, 4)).astype(np.float32)
            if rng.rand() < 0.2:
                init[..., 2] = rng.uniform(-1.5, 1.5, init[..., 2].shape)
            ref = co.rollout_cartpole(theta, init, E, T, obs_mask=0b1010 if pomdp else 0, physics64=p64)
            fit, ret, steps = es.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
            ok = np.array_equal(steps.cpu().numpy(), ref[2]) and np.array_equal(bits(fit.cpu().numpy()), bits(ref[0]))
            fit2 = es.rollout(dev(theta), dev(init), mode=mode)                 # the path without episode outputs
            ok = ok and np.array_equal(bits(fit2.cpu().numpy()), bits(ref[0]))
        elif kind in ("gru", "gru_mfma"):
            n, T = int(rng.choice([1, 5, 33, 130])), int(rng.choice([1, 9, 80]))
            E = int(rng.randint(12, 21)) if kind == "gru_mfma" else int(rng.randint(1, 12))
            pomdp = bool(rng.randint(0, 2))
            es = HipES("CartPole-v1", 4, 2, True, True, pomdp=pomdp, max_step=T, eval_ep_num=E)
            theta = (rng.randn(n, 6562) * min(sigma, 1.0)).astype(np.float32)
            init = rng.uniform(-0.05, 0.05, (E, 4) if shared else (n, E, 4)).astype(np.float32)
            ref = co.rollout_cartpole(theta, init, E, T, gru=True, obs_mask=0b1010 if pomdp else 0)
            fit, ret, steps = es.rollout(dev(theta), dev(init), mode=mode, want_episodes=True)
            ok = np.array_equal(steps.cpu().numpy(), ref[2]) and np.array_equal(bits(fit.cpu().numpy()), bits(ref[0]))
        elif kind in ("lander", "lander_mlp"):
            gru = kind == "lander"
            n, E, T = int(rng.choice([1, 6, 40])), int(rng.choice([1, 3, 5, 9, 13])), int(rng.choice([5, 60, 150]))
            pomdp = bool(rng.randint(0, 2))
            es = HipES("LunarLanderContinuous-v2", 8, 4, False, gru, pomdp=pomdp, max_step=T, eval_ep_num=E)
            es.set_tuning("box2d_lanes_per_env", int(rng.choice([0, 1, 2, 4, 8, 16, 64])))     # MLP kernel only
            theta = (rng.randn(n, es.P) * min(sigma, 1.0)).astype(np.float32)
            init = rng.uniform(0, 1, (E, 16) if shared else (n, E, 16)).astype(np.float32)
            ref = co.rollout_lander(theta, init, E, T, gru=gru, obs_mask=0b101100 if pomdp else 0)
            fit, ret, steps = es.rollout(dev(theta), dev(init), want_episodes=True)
            ok = (np.array_equal(steps.cpu().numpy(), ref[2]) and np.array_equal(bits(ret.cpu().numpy()), bits(ref[1])) and
                  np.array_equal(bits(fit.cpu().numpy()), bits(ref[0])))
        elif kind == "walker":
            n, E, T = int(rng.choice([1, 5, 33])), int(rng.choice([1, 2, 5])), int(rng.choice([5, 60, 150]))
            es = HipES("BipedalWalker-v3", 24, 4, False, False, max_step=T, eval_ep_num=E)
            es.set_tuning("box2d_lanes_per_env", int(rng.choice([0, 1, 2, 4, 8, 16, 64])))
            theta = (rng.randn(n, es.P) * sigma).astype(np.float32)
            init = rng.uniform(0, 1, (E, 4) if shared else (n, E, 4)).astype(np.float32)
            ref = co.rollout_walker(theta, init, E, T)
            fit, ret, steps = es.rollout(dev(theta), dev(init), want_episodes=True)
            ok = (np.array_equal(steps.cpu().numpy(), ref[2]) and np.array_equal(bits(ret.cpu().numpy()), bits(ref[1])) and
                  np.array_equal(bits(fit.cpu().numpy()), bits(ref[0])))
        else:
            na = int(rng.choice([2, 3]))
            n, E = int(rng.choice([1, 9, 100, 515])), int(rng.randint(1, 7))
            es = HipES("simple_spread", 6 * na, 5, True, False, max_step=25, eval_ep_num=E, n_agents=na)
            theta = (rng.randn(n, es.P) * sigma).astype(np.float32)
            init = rng.uniform(-1, 1, (E, 4 * na) if shared else (n, E, 4 * na)).astype(np.float32)
            ref = co.rollout_spread(theta, init, E, na)
            fit, ret, _ = es.rollout(dev(theta), dev(init), want_episodes=True)
            ok = np.array_equal(bits(ret.cpu().numpy()), bits(ref[1])) and np.array_equal(bits(fit.cpu().numpy()), bits(ref[0]))
        es.close()
        t = tally.setdefault(kind, [0, 0])
        t[0] += 1
        t[1] += int(ok)
        if not ok:
            print("MISMATCH", json.dumps({"case": case, "kind": kind, "n": n, "E": E, "mode": mode, "shared": shared, "sigma": sigma}))
    print(json.dumps({"cases": args.cases, "seed": args.seed, "by_kind": {k: {"cases": v[0], "bit_exact": v[1]} for k, v in tally.items()},
                      "all_bit_exact": all(v[0] == v[1] for v in tally.values())}))


if __name__ == "__main__":
    main()

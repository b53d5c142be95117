"""python sweep_main.py --cfg-path conf/cartpole_openai.yaml --init-sigma 0.3   (one trial, like the reference)
python sweep_main.py --sweep sweep_config/cartpole_openaies.yaml --count 16     (a whole sweep, wandb-free)

The reference's sweep_main.py (sweep_main.py:33-88) is a per-trial entry point that a `wandb agent` calls with
hyper-parameter flags; the sweep itself lives in the wandb service.  Here a generation takes milliseconds, so the
driver can run every trial of a sweep in this process.  The sweep files keep the reference's format
(sweep_config/*.yaml: method / metric / parameters with value | values | min,max).  `method: grid` enumerates the
`values` lists; `random` and `bayes` draw uniformly (there is no surrogate model here -- with trials this cheap a
larger random budget does the same job).  One line per trial goes to logs/sweeps/<name>/trials.jsonl and the best
trial is printed at the end.
"""
import argparse
import copy
import itertools
import json
import os
import random

import numpy as np
import torch
import yaml

import builder
from run_es import change_value, set_seed

_OVERRIDES = (("init_sigma", float), ("sigma_decay", float), ("learning_rate", float), ("elite_num", int),
              ("offspring_num", int))
_RUN_KEYS = {"cfg_path": str, "generation_num": int, "eval_ep_num": int, "seed": int}


def trial_points(spec, count, rng):
    """Yield {flag_name: value} dicts from a reference-format `parameters` block."""
    params = {k.replace("-", "_"): v for k, v in spec["parameters"].items()}
    fixed = {k: v["value"] for k, v in params.items() if "value" in v}
    lists = {k: v["values"] for k, v in params.items() if "values" in v}
    ranges = {k: (v["min"], v["max"]) for k, v in params.items() if "min" in v}
    if spec.get("method", "random") == "grid":
        if ranges:
            raise ValueError("grid sweeps need `values` lists, not min/max ranges: " + ", ".join(ranges))
        for n, combo in enumerate(itertools.product(*lists.values())):
            if count is not None and n >= count:
                return
            yield {**fixed, **dict(zip(lists.keys(), combo))}
        return
    for _ in range(count if count is not None else 10):
        point = dict(fixed)
        point.update({k: rng.choice(v) for k, v in lists.items()})
        for k, (lo, hi) in ranges.items():
            point[k] = rng.randint(lo, hi) if isinstance(lo, int) and isinstance(hi, int) else rng.uniform(lo, hi)
        yield point


def run_trial(config, point, args):
    """One ESLoop run with the point's overrides; returns the trial record."""
    config = copy.deepcopy(config)
    for key, typ in _OVERRIDES:
        if point.get(key) is not None:
            change_value(config, key, typ(point[key]))
    seed = int(point.get("seed", args.seed))
    config.setdefault("strategy", {}).setdefault("seed", seed)
    config.setdefault("env", {}).setdefault("seed", seed)
    set_seed(seed)
    loop = builder.build_loop(config, int(point.get("generation_num", args.generation_num)), args.process_num,
                              int(point.get("eval_ep_num", args.eval_ep_num)), args.log, args.save_model_period)
    loop.run()
    best = [b for b, _ in loop.history]
    tail = best[-5:]
    return {"ep5_mean_reward": sum(tail) / max(len(tail), 1), "best_reward": max(best) if best else None,
            "generations": len(best), "log_dir": loop.save_dir}


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--cfg-path", type=str, default="conf/lunarlander_openai.yaml", help="config file to run.")
    parser.add_argument("--seed", type=int, default=0, help="random seed.")
    parser.add_argument("--process-num", type=int, default=12, help="kept for compatibility (no worker processes).")
    parser.add_argument("--generation-num", type=int, default=1000, help="max number of generation iteration.")
    parser.add_argument("--eval-ep-num", type=int, default=5, help="number of model evaluaion per iteration.")
    parser.add_argument("--log", action="store_true", help="wandb log")
    parser.add_argument("--save-model-period", type=int, default=10, help="save model for every n iteration.")
    for key, typ in _OVERRIDES:
        parser.add_argument("--" + key.replace("_", "-"), type=typ, default=None)
    parser.add_argument("--sweep", type=str, default=None, help="sweep file (reference sweep_config format).")
    parser.add_argument("--count", type=int, default=None, help="number of trials (default: whole grid / 10 draws).")
    args = parser.parse_args()

    if args.sweep is None:                              # the reference's behaviour: one trial from the flags
        with open(args.cfg_path) as f:
            config = yaml.load(f, Loader=yaml.FullLoader)
        record = run_trial(config, {k: getattr(args, k) for k, _ in _OVERRIDES}, args)
        print(json.dumps(record))
        return

    with open(args.sweep) as f:
        spec = yaml.load(f, Loader=yaml.FullLoader)
    metric = (spec.get("metric") or {}).get("name", "ep5_mean_reward")
    sign = -1.0 if (spec.get("metric") or {}).get("goal", "maximize") == "minimize" else 1.0
    name = os.path.splitext(os.path.basename(args.sweep))[0]
    out_dir = os.path.join("logs", "sweeps", name)
    os.makedirs(out_dir, exist_ok=True)
    rng = random.Random(args.seed)
    best = None
    for n, point in enumerate(trial_points(spec, args.count, rng)):
        cfg_path = point.get("cfg_path", args.cfg_path)
        with open(cfg_path) as f:
            config = yaml.load(f, Loader=yaml.FullLoader)
        record = {"trial": n, "params": point, **run_trial(config, point, args)}
        with open(os.path.join(out_dir, "trials.jsonl"), "a") as f:
            f.write(json.dumps(record) + "\n")
        print(f"trial {n}: {metric} = {record[metric]:.3f}  params = {point}")
        if best is None or sign * record[metric] > sign * best[metric]:
            best = record
    print("best:", json.dumps(best))


if __name__ == "__main__":
    main()

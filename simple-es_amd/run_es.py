"""python run_es.py --cfg-path conf/cartpole.yaml  -- same flags as the reference CLI (run_es.py:15-62).

Multi-GPU: launch one rank per GPU with torch.distributed.run; the population is sharded over the ranks
and the fitness vector is all-gathered inside the library (peer stores over xGMI, RCCL as the second transport).
"""
import argparse
import os
import random

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC for the multi-GPU transports; before any HIP call

import numpy as np
import torch
import yaml

import builder


def set_seed(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def change_value(tree, key, value):
    """set every occurrence of `key` in a nested config dict (what sweep_main.py:16-30 does for wandb sweeps)"""
    for k, v in tree.items():
        if k == key:
            tree[k] = value
        elif isinstance(v, dict):
            change_value(v, key, value)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--cfg-path", type=str, default="conf/cartpole.yaml", help="config file to run.")
    parser.add_argument("--seed", type=int, default=0, help="random seed.")
    parser.add_argument("--process-num", type=int, default=12,
                        help="kept for compatibility: the device rollout has no worker processes.")
    parser.add_argument("--generation-num", type=int, default=10000, help="max number of generation iteration.")
    parser.add_argument("--eval-ep-num", type=int, default=5, help="number of model evaluaion per iteration.")
    parser.add_argument("--log", action="store_true", help="wandb log")
    parser.add_argument("--save-model-period", type=int, default=10, help="save model for every n iteration.")
    # hyper-parameter overrides, the flag set of the reference's sweep driver (sweep_main.py:65-69); unset = keep YAML
    for flag, typ in (("--init-sigma", float), ("--sigma-decay", float), ("--learning-rate", float),
                      ("--elite-num", int), ("--offspring-num", int)):
        parser.add_argument(flag, type=typ, default=None, help="override the strategy value of the config file.")
    args = parser.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if os.environ.get("SES_DIST_BACKEND", "nccl") == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:                           # test rigs where the ranks share one GPU (RCCL refuses duplicate devices): gloo control plane,
            dist.init_process_group(os.environ["SES_DIST_BACKEND"])      # the fitness exchange stays the library's peer stores

    set_seed(args.seed)
    with open(args.cfg_path) as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    for key in ("init_sigma", "sigma_decay", "learning_rate", "elite_num", "offspring_num"):
        if getattr(args, key) is not None:
            change_value(config, key, getattr(args, key))
    config.setdefault("strategy", {}).setdefault("seed", args.seed)
    config.setdefault("env", {}).setdefault("seed", args.seed)

    loop = builder.build_loop(config, args.generation_num, args.process_num, args.eval_ep_num, args.log,
                              args.save_model_period)
    loop.run()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

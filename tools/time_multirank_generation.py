#!/usr/bin/env python3
"""Time per generation of the device-side loop (ESLoop.generations -> ses_run_generations) with the ranks SHARING one GPU and a
short rollout (max_step 20), so that what a generation costs besides the rollout -- episode mean, the two exchanges, the tail --
is most of it and a launch saved per rank shows.  Run once per setting of SES_TUNING (fused_fitness_exchange=0,
openai_granule_exchange=0) and compare:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/time_multirank_generation.py [rows per rank]
The ranks time-slice the one device, so a generation is roughly the SUM of the ranks' kernels: differences count per rank."""
import json, os, sys, statistics, tempfile, time
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "simple-es_amd")]
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
per = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
import builder
os.chdir(tempfile.mkdtemp(prefix="ses_tmg_"))
cfg = {"env": {"name": "CartPole-v1", "max_step": 20, "pomdp": False, "seed": 0, "shared_init": True, "fixed_length": True},
       "network": {"name": "gym_model", "num_state": 4, "num_action": 2, "discrete_action": True, "gru": False},
       "strategy": {"name": "openai_es", "init_sigma": 0.1, "sigma_decay": 0.999, "learning_rate": 0.05, "offspring_num": per * world, "seed": 0}}
loop = builder.build_loop(cfg, 0, 1, 5, False, 10 ** 9)
pop = loop.offspring_strategy.init_offspring(loop.network, loop.env.get_agent_ids())
pop = loop.generations(pop, 200)
torch.cuda.synchronize(); dist.barrier()
ts = []
for rep in range(9):
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    pop = loop.generations(pop, 400)
    torch.cuda.synchronize(); dist.barrier()
    ts.append((time.perf_counter() - t0) / 400 * 1e6)
owner = getattr(loop.dev, "_comm_owner", None)
if rank == 0:
    print(json.dumps({"ranks_on_one_gpu": world, "rows_per_rank": per, "tuning": os.environ.get("SES_TUNING", ""),
                      "us_per_generation": round(statistics.median(ts), 2), "min": round(min(ts), 2),
                      "exchanges_flag_granule": owner.comm_p2p_counts() if owner else None,
                      "device_loop": bool(loop.batched_generations or loop.device_side_loop)}), flush=True)
dist.destroy_process_group()

#!/usr/bin/env python3
"""Host time to ENQUEUE one generation of the product loop (ESLoop.generation, openai_es, 4096 offspring) next to the
GPU time it takes: if the first is not well below the second, the loop is launch-bound on the host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
import torch, tempfile
os.chdir(tempfile.mkdtemp())
import builder
fused = (sys.argv[1] != "0") if len(sys.argv) > 1 else True
cfg = {"env": {"name": "CartPole-v1", "max_step": 500, "pomdp": False, "seed": 0, "shared_init": True, "fixed_length": True},
       "network": {"name": "gym_model", "num_state": 4, "num_action": 2, "discrete_action": True, "gru": False},
       "strategy": {"name": "openai_es", "init_sigma": 0.1, "sigma_decay": 0.999, "learning_rate": 0.05, "offspring_num": 4096, "seed": 0}}
loop = builder.build_loop(cfg, 0, 1, 5, False, 10 ** 9)
loop.offspring_strategy.fused = fused
pop = loop.offspring_strategy.init_offspring(loop.network, loop.env.get_agent_ids())
for _ in range(300):
    pop, *_ = loop.generation(pop)
torch.cuda.synchronize()
K = 500
t0 = time.perf_counter()
for _ in range(K):
    pop, *_ = loop.generation(pop)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"fused={fused}: host enqueue {1e6*(t1-t0)/K:.1f} us / generation; until the GPU is done {1e6*(t2-t0)/K:.1f} us / generation")

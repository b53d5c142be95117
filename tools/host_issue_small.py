#!/usr/bin/env python3
"""Host time to ENQUEUE one generation against the time until the GPU is done, for the reference's small configs
(conf/cartpole.yaml: simple_evolution, 96 offspring; conf/cartpole_pomdp_gru.yaml; conf/simplespread.yaml): below ~1000
offspring a generation is a few dependent kernels of a lone wave each, and the question is whether the Python host keeps up."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
import torch, yaml
os.chdir(tempfile.mkdtemp())
import builder
for name in sys.argv[1:] or ["cartpole.yaml", "cartpole_openai.yaml", "cartpole_pomdp_gru.yaml", "simplespread.yaml"]:
    cfg = yaml.load(open(os.path.join(ROOT, "simple-es_amd", "conf", name)), Loader=yaml.FullLoader)
    loop = builder.build_loop(cfg, 0, 1, 5, False, 10 ** 9)
    pop = loop.offspring_strategy.init_offspring(loop.network, loop.env.get_agent_ids())
    for _ in range(200):
        pop, *_ = loop.generation(pop)
    torch.cuda.synchronize()
    K = 500
    t0 = time.perf_counter()
    for _ in range(K):
        pop, *_ = loop.generation(pop)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e6*(t1-t0)/K:.1f} us / generation; until the GPU is done {1e6*(t2-t0)/K:.1f} us / generation", flush=True)

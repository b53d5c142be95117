import os, sys, time, tempfile, cProfile, pstats, io
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
import torch, yaml
os.chdir(tempfile.mkdtemp())
import builder
name = sys.argv[1]
cfg = yaml.load(open(os.path.join(ROOT, "simple-es_amd", "conf", name)), Loader=yaml.FullLoader)
loop = builder.build_loop(cfg, 0, 1, 5, False, 10 ** 9)
pop = loop.offspring_strategy.init_offspring(loop.network, loop.env.get_agent_ids())
for _ in range(200):
    pop, *_ = loop.generation(pop)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(1000):
    pop, *_ = loop.generation(pop)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])

"""Wall time per generation of ESLoop.run() for a config of simple-es_amd/conf (development helper).
usage: time_loop.py <config.yaml> [offspring_num | 0 = the config's] [generations = 1000] [save_model_period = 100000]
(run_es.py's default checkpoint period is 10)"""
import os, sys, time, io, contextlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
os.chdir("/tmp")
import yaml, torch
import builder
cfg = yaml.load(open(os.path.join(ROOT, "simple-es_amd", "conf", sys.argv[1])), Loader=yaml.FullLoader)
if len(sys.argv) > 2 and int(sys.argv[2]) > 0:
    cfg["strategy"]["offspring_num"] = int(sys.argv[2])
gens = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
period = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
loop = builder.build_loop(cfg, gens, 1, 5, False, period)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()      # HIP context / first allocation is start-up, not loop time
buf = io.StringIO()
t0 = time.perf_counter()
with contextlib.redirect_stdout(buf):
    loop.run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(sys.argv[1:], "%d generations in %.3f s -> %.3f ms per generation; last best %.1f" % (gens, dt, 1e3 * dt / gens, loop.history[-1][0]))

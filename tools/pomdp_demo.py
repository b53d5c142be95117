#!/usr/bin/env python3
"""Development demo: GRU vs MLP policy on POMDP CartPole (the reference's only published learning result,
README.md:42: GRU + simple_evolution reaches 500, the MLP stays around 60)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "simple-es_amd")]
import yaml
import builder

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 150
os.chdir(tempfile.mkdtemp())
for gru in (True, False):
    cfg = yaml.load(open(os.path.join(ROOT, "simple-es_amd", "conf", "cartpole_pomdp_gru.yaml")), Loader=yaml.FullLoader)
    cfg["network"]["gru"] = gru
    loop = builder.build_loop(cfg, gens, 1, 5, False, 10 ** 9)
    t0 = time.time()
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        loop.run()
    best = [b for b, _ in loop.history]
    print(f"gru={gru}: {gens} generations in {time.time() - t0:.1f}s; best per 10 generations:",
          [round(max(best[i:i + 10]), 1) for i in range(0, gens, 10)])

#!/usr/bin/env python3
"""cProfile of ESLoop.run() (prints, metrics.jsonl, read-backs included) for one config: where the host time of a
generation goes when the GPU is not the bottleneck.  usage: profile_run.py <config.yaml> [generations]"""
import os, sys, tempfile, cProfile, pstats, io, contextlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
import torch, yaml
os.chdir(tempfile.mkdtemp())
import builder
cfg = yaml.load(open(os.path.join(ROOT, "simple-es_amd", "conf", sys.argv[1])), Loader=yaml.FullLoader)
gens = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
loop = builder.build_loop(cfg, gens, 1, 5, False, 10 ** 9)
pr = cProfile.Profile()
with open(os.devnull, "w") as sink, contextlib.redirect_stdout(sink):
    pr.enable(); loop.run(); pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:4000])

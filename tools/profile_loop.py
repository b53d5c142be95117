#!/usr/bin/env python3
"""cProfile of the host side of ESLoop.run() (development helper): python tools/profile_loop.py cartpole.yaml"""
import contextlib, cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
os.chdir("/tmp")
import yaml
import builder
cfg = yaml.load(open(os.path.join(ROOT, "simple-es_amd", "conf", sys.argv[1])), Loader=yaml.FullLoader)
loop = builder.build_loop(cfg, 500, 1, 5, False, 100000)
pr = cProfile.Profile()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable()
    loop.run()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)

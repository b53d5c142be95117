import os, sys, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "simple-es_amd"))
import torch, bench
from ses import HipES
es = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=5)
keep = []
for i in range(6):
    r = bench.env_step_roofline(es, 1 << 24)
    print(round(r["avg_launch_us"], 2), round(r["frac"], 4), flush=True)
    if i % 2 == 0:
        keep.append(torch.empty(1 << 26, device="cuda"))     # perturb the allocator between measurements

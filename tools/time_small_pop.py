import json, os, sys, statistics
import torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "simple-es_amd"))
from ses import HipES, MODE_FIXED_LENGTH
es = HipES("CartPole-v1", 4, 2, True, False, max_step=500, eval_ep_num=5)
for n in (96, 512, 1024, 4096):
    mu = es.zeros(es.P); theta = es.perturb(mu, 0.1, 0, 0, 0, n); init = es.init_states_uniform(0,0,0,n); fit = es.empty(n)
    for _ in range(5): es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit)
    torch.cuda.synchronize(); ts=[]
    for _ in range(9):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit)
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1)/20*1e3)
    print(n, round(statistics.median(ts),1), "us")

import json, os, sys, statistics
import torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "simple-es_amd"))
from ses import HipES
for n in (4096, 2048, 8192):
    for g in (1, 2, 4):
        es = HipES("LunarLanderContinuous-v2", 8, 4, False, True, pomdp=True, max_step=300, eval_ep_num=5)
        es.set_tuning("gru_ep_parallel_max", 0); es.set_tuning("lander_offspring_per_wave", g)
        mu = es.zeros(es.P); theta = es.perturb(mu, 0.168, 0, 0, 0, n); init = es.init_states_uniform(0,0,0,n); fit = es.empty(n)
        es.rollout(theta, init, fitness=fit); torch.cuda.synchronize(); ts=[]
        for _ in range(3):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record(); es.rollout(theta, init, fitness=fit); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        print(n, "offspring_per_wave", g, round(statistics.median(ts),2), "ms", flush=True)
        es.close()

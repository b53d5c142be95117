#!/bin/bash
# round 6, call H: the packed contraction of the sequential / episode-parallel GRU slice (ses_gru.h) -- parity on every GRU path, then timings
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_gru.py tests/test_gpu_lander.py tests/test_gpu_g9.py tests/test_gpu_learning.py -x -q > gpurun_out/r06_pytest_h.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r06_pytest_h.log
[ $rc = 0 ] || exit $rc
for lib in simple-es_amd/libses_prev.so simple-es_amd/libses_hip.so; do
echo "== $lib"
export SES_LIB_PATH=$R/$lib
python tools/time_loop.py cartpole_pomdp_gru.yaml 2>&1 | tail -1
python tools/time_loop.py lunarlander_openai.yaml 0 200 2>&1 | tail -1
python - <<'PY'
import os, sys, json, statistics, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "simple-es_amd")]
from ses import HipES, MODE_FIXED_LENGTH
for n, knobs, label in ((97, {}, "97 offspring, episode-parallel (lone waves)"), (800, {}, "800 offspring, episode-parallel (4 waves per SIMD)"),
                        (4096, {"gru_sequential": 1}, "4096 offspring, sequential kernel")):
    es = HipES("CartPole-v1", 4, 2, True, True, pomdp=True, max_step=500, eval_ep_num=5)
    for k, v in knobs.items(): es.set_tuning(k, v)
    theta = es.perturb(es.zeros(es.P), 0.1, 0, 0, 0, n)
    init = es.init_states_uniform(0, 0, 0, 1, shared=True)[0].contiguous()
    fit = es.empty(n)
    for _ in range(3): es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): es.rollout(theta, init, mode=MODE_FIXED_LENGTH, fitness=fit)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    print(json.dumps({"case": label, "rollout_ms": round(statistics.median(ts), 4)}), flush=True)
PY
done

#!/bin/bash
# round 4, second GPU call: how many hardware queues the in-process 8-rank rig needs; bench rehearsals on the gloo rig
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for q in 4 8 9 16 24; do
  echo "GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q SES_TAIL_SHAPES=8x4096 timeout -k 10 120 python tools/time_tail.py 2>&1 | tail -2
done | tee gpurun_out/hwq.txt
SES_TAIL_SHAPES=4x8192,4x16384 timeout -k 10 120 python tools/time_tail.py 2>&1 | tail -2 | tee gpurun_out/time_tail_w4.txt
for n in 2 4; do
  SES_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus $n --steps 100 --warmup 20 --blocks 9 --no-roofline --no-cpu-baseline --loop-generations 300 > gpurun_out/bench_gloo_$n.json 2> gpurun_out/bench_gloo_$n.err
  echo "bench $n rc=$?"; python - <<PY
import json
d=json.loads(open("gpurun_out/bench_gloo_$n.json").read().strip().splitlines()[-1])
for k in ("weak_4096_per_gpu","strong_4096_total","c4_65536_total"):
    v=d.get(k,{})
    print(k, {x: (round(v[x],2) if isinstance(v.get(x),float) else v.get(x)) for x in ("ms_per_step","rollout_us","allgather_us","fitness_loop_us","allgather_transport","error")})
print("timed_call", d["config"]["timed_call"][:60], "loop_ms", d.get("loop_ms_per_generation"), d.get("loop_error"))
PY
done

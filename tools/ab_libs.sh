#!/bin/bash
# development helper: A/B two library builds over population sizes (default split, i.e. what ses_rollout picks)
mkdir -p gpurun_out
out=gpurun_out/ab.log; : > $out
for lib in "$@"; do
  for n in 1024 2048 4096 8192 16384 65536; do
    SES_LIB_PATH=$PWD/$lib python bench.py --steps 200 --warmup 50 --offspring-per-gpu $n --no-cpu-baseline --no-roofline 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$lib','n',$n,'ms_per_gen',round(d['ms_per_step'],4),'rollout_ms',round(d['rollout_kernel']['ms'],4),'%.3e'%d['rollout_kernel']['env_steps_per_s_one_gpu'])" >> $out
  done
done
cat $out

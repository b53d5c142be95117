#!/bin/bash
# development helper: LPE x population sweep of the fused rollout for one or more library builds
mkdir -p gpurun_out
out=gpurun_out/lpe.log; : > $out
for lib in "$@"; do
  for l in 1 2 4 8; do for n in 4096 65536; do
    SES_LIB_PATH=$PWD/$lib python bench.py --steps 5 --warmup 2 --lanes-per-env $l --offspring-per-gpu $n --no-cpu-baseline --no-roofline 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$lib','lpe',$l,'n',$n,'ms_per_gen',round(d['ms_per_step'],3),'rollout_ms',round(d['rollout_kernel']['ms'],3),'%.3e'%d['rollout_kernel']['env_steps_per_s_one_gpu'])" >> $out
  done; done
done
cat $out

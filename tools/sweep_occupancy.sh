#!/bin/bash
# development helper: rollout time vs waves per SIMD for the pure LPE kernels (1024 SIMDs on the chip)
mkdir -p gpurun_out
out=gpurun_out/occupancy.log; : > $out
for l in 8 4; do
  for waves in 512 1024 1536 2048 2560 3072 4096; do
    n=$(( waves * 64 / l / 5 ))
    for blk in 64 256; do
      SES_TUNING=rollout_block=$blk python bench.py --steps 5 --warmup 2 --lanes-per-env $l --offspring-per-gpu $n --no-cpu-baseline --no-roofline 2>/dev/null \
        | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('lpe',$l,'block',$blk,'waves',$waves,'n',$n,'rollout_ms',round(d['rollout_kernel']['ms'],4))" >> $out
    done
  done
done
cat $out

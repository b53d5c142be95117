#!/bin/bash
# development helper: A/B library builds on the headline generation loop (long timed region, alternating order)
for rep in 1 2 3; do
  for lib in "$@"; do
    SES_LIB_PATH=$PWD/$lib python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$lib','ms_per_gen',round(d['ms_per_step'],4),'rollout_ms',round(d['rollout_kernel']['ms'],4))"
  done
done

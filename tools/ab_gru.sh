#!/bin/bash
# development helper: A/B library builds on the GRU bench workload
for lib in "$@"; do
  SES_LIB_PATH=$PWD/$lib python bench.py --gru --no-cpu-baseline --no-roofline 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$lib','ms_per_gen',round(d['ms_per_step'],4),'rollout_ms',round(d['rollout_kernel']['ms'],4))"
done

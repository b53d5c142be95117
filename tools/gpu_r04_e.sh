#!/bin/bash
# round 4: the sharded tail by kernel (granule path and float all-gather path) at the 8-rank shapes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for mode in 1 0; do
  rm -rf gpurun_out/prof_tail
  (cd /tmp && export TMPDIR=/tmp && SES_TUNING=openai_granule_exchange=$mode SES_TAIL_SHAPES=8x4096,8x8192 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tail -- python3 $R/tools/time_tail.py > $R/gpurun_out/prof_tail_$mode.txt 2>&1)
  echo "== openai_granule_exchange=$mode"
  python tools/tail_by_kernel.py $(find gpurun_out/prof_tail -name "*kernel_trace.csv" | head -1)
done > gpurun_out/r04_tail_by_kernel.txt 2>&1
rm -rf gpurun_out/prof_tail
cat gpurun_out/r04_tail_by_kernel.txt

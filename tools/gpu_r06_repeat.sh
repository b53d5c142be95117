#!/bin/bash
# round 6: every multi-rank GPU test six times in a row on one box (no retry anywhere), then the in-process rig twenty times
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
out=gpurun_out/r06_multirank_repeat.txt; : > $out
for i in 1 2 3 4 5 6; do
  r=$(timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_device_loop_world8.py tests/test_gpu_sharded_tail.py tests/test_gpu_comm.py -q -m gpu 2>&1 | tail -1)
  echo "run $i: $r" | tee -a $out
  case "$r" in *failed*|*error*) echo FAILED | tee -a $out; exit 1;; esac
done
echo "all 6 runs passed" | tee -a $out

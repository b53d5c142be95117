#!/bin/bash
# The in-process many-rank rig (tests/test_gpu_sharded_tail.py), N times in a row on one box: every parametrisation must pass
# every time, there is no retry.  Usage (gpurun): bash tools/sharded_tail_repeat.sh [N=20] > gpurun_out/sharded_tail_repeat.txt
N="${1:-20}"
fail=0
for i in $(seq 1 "$N"); do
  out=$(timeout -k 10 600 python -m pytest tests/test_gpu_sharded_tail.py -q -m gpu -x 2>&1 | tail -1)
  echo "run $i: $out"
  case "$out" in *failed*|*error*) fail=1; break;; esac
done
[ "$fail" = 0 ] && echo "all $N runs passed" || { echo "FAILED"; exit 1; }

#!/bin/bash
# round 6, call B: parity of the reworked 16-lane fc2 and the new 32-lanes-per-env split, the bench self-certification tests, and the A/B of
# the small per-GPU populations (profiles/r06_small_populations.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_g9.py tests/test_gpu_multirank.py -x -q -k "rollout or g9_cartpole or bench" > gpurun_out/r06_pytest_b.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/r06_pytest_b.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/fuzz_parity.py --cases 1500 --seed 61 --only mlp > gpurun_out/r06_fuzz_mlp.txt 2>&1; rc=$?
echo "fuzz rc=$rc"; tail -3 gpurun_out/r06_fuzz_mlp.txt
[ $rc = 0 ] || exit $rc
timeout -k 10 400 python tools/time_small_populations.py > gpurun_out/r06_small_populations.txt 2>&1; rc=$?
SES_TAIL_SHAPES=1x4096,2x2048,4x1024,8x512 timeout -k 10 120 python tools/time_tail.py > gpurun_out/r06_time_tail_strong.txt 2>&1; cat gpurun_out/r06_time_tail_strong.txt
echo "small rc=$rc"; cat gpurun_out/r06_small_populations.txt

#!/bin/bash
# round 6, call D: the (8, 16) mixed split and the 4x4x1 MFMA GRU step -- parity, then timings
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gru.py -x -q -k "mixed_splits or 4x4x1 or rollout_golden or mfma_4x4x1" > gpurun_out/r06_pytest_d.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -15 gpurun_out/r06_pytest_d.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/time_gru.py > gpurun_out/r06_time_gru.txt 2>&1; cat gpurun_out/r06_time_gru.txt
timeout -k 10 400 python tools/time_small_populations.py > gpurun_out/r06_small_populations.txt 2>&1; cat gpurun_out/r06_small_populations.txt

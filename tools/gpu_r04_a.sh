#!/bin/bash
# round 4, first GPU call: the sharded tail + device-side multi-rank loop tests, the tail timings, a 2-rank bench rehearsal
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_tail.py tests/test_gpu_multirank.py tests/test_gpu_host_mirror.py tests/test_gpu_comm.py -x -q > gpurun_out/pytest_a.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest_a.log
timeout -k 10 300 python tools/time_tail.py > gpurun_out/time_tail.txt 2> gpurun_out/time_tail.err; echo "time_tail rc=$?"; cat gpurun_out/time_tail.txt; tail -3 gpurun_out/time_tail.err

#!/bin/bash
# round 4, fourth GPU call: the granule exchange of the sharded tail -- tests, tail timings
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_tail.py tests/test_gpu_multirank.py -x -q > gpurun_out/pytest_d.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_d.log
SES_TAIL_SHAPES=2x4096,4x4096,4x8192,4x16384 timeout -k 10 300 python tools/time_tail.py 2>&1 | tail -4 | tee gpurun_out/time_tail_d.txt
SES_TUNING=openai_granule_exchange=0 SES_TAIL_SHAPES=4x4096,4x8192 timeout -k 10 300 python tools/time_tail.py 2>&1 | tail -2 | tee -a gpurun_out/time_tail_d.txt

#!/bin/bash
# round 6 (evidence at depth, ~17 minutes): skip reasons of the real-device tests, a long fuzz pass, the soak runs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_comm.py -q -rs > gpurun_out/r06_pytest_comm_skips.log 2>&1; tail -12 gpurun_out/r06_pytest_comm_skips.log
rm -f gpurun_out/fuzz_[0-9]*.txt
bash tools/fuzz_round.sh ${1:-420} ${2:-900} > gpurun_out/r06_fuzz_parity_long.txt 2>&1; tail -1 gpurun_out/r06_fuzz_parity_long.txt | cut -c1-400
grep -h MISMATCH gpurun_out/fuzz_[0-9]*.txt | head -5
bash tools/soak.sh > /dev/null 2>&1; cp gpurun_out/soak.txt gpurun_out/r06_soak.txt; cat gpurun_out/r06_soak.txt
rm -f gpurun_out/fuzz_[0-9]*.txt

#!/bin/bash
# round 6, call F: after the alignment fix in ses_gru_mfma4.h -- its tests and timing again, and the SQ profile of the MFMA GRU kernels
# (the hash fragment k_rollout_gru_mfma covers both kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_gru.py -x -q -k "4x4x1 or matrix_cores" > gpurun_out/r06_pytest_f.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r06_pytest_f.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/time_gru.py > gpurun_out/r06_time_gru.txt 2>&1; cat gpurun_out/r06_time_gru.txt
bash tools/prof_mfma.sh > gpurun_out/sq_mfma.txt 2>&1
python tools/collect_sq.py r06 gru_mfma k_rollout_gru_mfma gpurun_out/mf_1 gpurun_out/mf_2
cp profiles/r06_sq_gru_mfma.json gpurun_out/
rm -rf gpurun_out/mf_1 gpurun_out/mf_2

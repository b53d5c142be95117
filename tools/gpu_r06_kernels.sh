#!/bin/bash
# round 6: the short gpurun call behind the kernel work of the round (a few minutes) -- parity of every CartPole MLP split (scalar and
# packed, 32 lanes, mixed) and of every GRU path incl. the 4x4x1 MFMA step, MLP fuzz, then the timings quoted in DESIGN.md section 4:
# small per-GPU populations, the tail of the strong line by rank count, the GRU rollout by kernel; finally the SQ profile of the MFMA
# GRU kernels (hash fragment k_rollout_gru_mfma).  Outputs under gpurun_out/ (copy what is to be kept into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_g9.py tests/test_gpu_gru.py -x -q -k "rollout or g9_cartpole or 4x4x1 or matrix_cores" > gpurun_out/r06_pytest_kernels.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r06_pytest_kernels.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/fuzz_parity.py --cases 1500 --seed 61 --only mlp > gpurun_out/r06_fuzz_mlp.txt 2>&1; rc=$?
echo "fuzz rc=$rc"; tail -1 gpurun_out/r06_fuzz_mlp.txt | cut -c1-200
[ $rc = 0 ] || exit $rc
timeout -k 10 400 python tools/time_small_populations.py > gpurun_out/r06_small_populations.txt 2>&1; cat gpurun_out/r06_small_populations.txt
SES_TAIL_SHAPES=1x4096,2x2048,4x1024,8x512 timeout -k 10 120 python tools/time_tail.py > gpurun_out/r06_time_tail_strong.txt 2>&1; cat gpurun_out/r06_time_tail_strong.txt
timeout -k 10 300 python tools/time_gru.py > gpurun_out/r06_time_gru.txt 2>&1; cat gpurun_out/r06_time_gru.txt
bash tools/prof_mfma.sh > gpurun_out/sq_mfma.txt 2>&1
python tools/collect_sq.py r06 gru_mfma k_rollout_gru_mfma gpurun_out/mf_1 gpurun_out/mf_2
cp profiles/r06_sq_gru_mfma.json gpurun_out/
rm -rf gpurun_out/mf_1 gpurun_out/mf_2

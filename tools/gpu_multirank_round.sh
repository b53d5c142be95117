#!/bin/bash
# the multi-rank round (one gpurun call): multi-rank tests (skipped with "notests"), exchange latency by kind, a generation of two ranks with
# and without the fused exchanges, the openai_es tail replicated / in shard form, the 8-rank shapes by kernel from a trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
if [ "$1" != "notests" ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_sharded_tail.py tests/test_gpu_multirank.py tests/test_gpu_comm.py -x -q > gpurun_out/pytest_f.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_f.log
fi
for w in 2 4; do
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500 + w)) tools/time_allgather.py 2>/dev/null | grep "^{"
done | tee gpurun_out/r06_time_allgather.txt
for tuning in "" "fused_fitness_exchange=0" "openai_granule_exchange=0" "fused_fitness_exchange=0,openai_granule_exchange=0"; do
  SES_TUNING=$tuning timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/time_multirank_generation.py 8192 2>/dev/null | grep "^{"
done | tee gpurun_out/r06_time_multirank_generation.txt
SES_TAIL_SHAPES=2x4096,4x4096,4x8192,4x16384 timeout -k 10 300 python tools/time_tail.py 2>&1 | grep "^{" | tee gpurun_out/r06_time_tail_sharded.txt
echo "float all-gather of the partials:" | tee -a gpurun_out/r06_time_tail_sharded.txt
SES_TUNING=openai_granule_exchange=0 SES_TAIL_SHAPES=4x4096,4x8192 timeout -k 10 300 python tools/time_tail.py 2>&1 | grep "^{" | tee -a gpurun_out/r06_time_tail_sharded.txt
for mode in 1 0; do
  rm -rf gpurun_out/prof_tail
  (cd /tmp && export TMPDIR=/tmp && SES_TUNING=openai_granule_exchange=$mode SES_TAIL_SHAPES=8x4096,8x8192 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tail -- python3 $R/tools/time_tail.py > $R/gpurun_out/prof_tail_$mode.txt 2>&1; echo "rocprofv3 exit code $?")
  echo "== openai_granule_exchange=$mode"
  python tools/tail_by_kernel.py $(find gpurun_out/prof_tail -name "*kernel_trace.csv" | head -1)
done > gpurun_out/r06_tail_by_kernel.txt 2>&1
rm -rf gpurun_out/prof_tail
cat gpurun_out/r06_tail_by_kernel.txt

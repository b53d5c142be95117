#!/bin/bash
# round 6, call A: the new multi-rank tests, the bench line at N = 1, and the gloo rehearsals of the N > 1 line (ranks share the GPU)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
set -o pipefail
[ -n "$SKIP_TESTS" ] && true || timeout -k 10 600 python -m pytest tests/test_gpu_comm.py -x -q > gpurun_out/r06_pytest_comm.log 2>&1; rc=$?; echo "pytest comm rc=$rc"; tail -5 gpurun_out/r06_pytest_comm.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; rc=$?; echo "bench n1 rc=$rc"; tail -3 gpurun_out/r06_bench_n1.err
[ $rc = 0 ] || exit $rc
for n in 2 4; do
  SES_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus $n --steps 20 --warmup 5 > gpurun_out/r06_bench_gloo$n.json 2> gpurun_out/r06_bench_gloo$n.err; rc=$?
  echo "bench gloo $n rc=$rc"; tail -3 gpurun_out/r06_bench_gloo$n.err
  [ $rc = 0 ] || exit $rc
done

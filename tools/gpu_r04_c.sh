#!/bin/bash
# round 4, third GPU call: launch-shape tests, the sharded tail by kernel, the tanh-table conflict A/B, the SLP build of the Box2D step
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "env_step" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_c.log
rm -rf gpurun_out/prof_tail
(cd /tmp && export TMPDIR=/tmp && SES_TAIL_SHAPES=8x4096,8x8192 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tail -- python3 $R/tools/time_tail.py > $R/gpurun_out/prof_tail.txt 2>&1)
python tools/tail_by_kernel.py $(find gpurun_out/prof_tail -name "*kernel_trace.csv" | head -1) > gpurun_out/r04_tail_by_kernel.txt 2>&1; cat gpurun_out/r04_tail_by_kernel.txt
rm -rf gpurun_out/prof_tail
timeout -k 10 600 bash tools/ab_tanh_conflicts.sh 2>&1 | tee gpurun_out/r04_ab_tanh_conflicts.txt
rm -rf gpurun_out/sq_tc
for rep in 1 2; do
  for lib in simple-es_amd/libses_hip.so ab/libSLP.so; do
    echo "== $lib"
    SES_LIB_PATH=$R/$lib timeout -k 10 200 python tools/lander_step_cost.py 2>/dev/null | head -2 | cut -c1-200
    SES_LIB_PATH=$R/$lib timeout -k 10 200 python tools/time_box2d_mlp.py 4096 0 2>/dev/null | cut -c1-140
    SES_LIB_PATH=$R/$lib timeout -k 10 200 python tools/time_c3.py 2>/dev/null | tail -2 | cut -c1-200
  done
done 2>&1 | tee gpurun_out/r04_ab_slp.txt

#!/bin/bash
# One gpurun call of the usual round-trip: GPU tests, the bench line, loop timings.  Output under gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
python bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -c 6000 gpurun_out/bench.json
for cfg in cartpole_openai.yaml cartpole.yaml cartpole_pomdp_gru.yaml simplespread.yaml lunarlander_openai.yaml; do
  python tools/time_loop.py $cfg 2>&1 | tail -1
done | tee gpurun_out/time_loop.txt
python tools/time_loop.py lunarlander.yaml 0 300 2>&1 | tail -1 | tee -a gpurun_out/time_loop.txt
python tools/time_loop.py bipedalwalker.yaml 0 60 2>&1 | tail -1 | tee -a gpurun_out/time_loop.txt

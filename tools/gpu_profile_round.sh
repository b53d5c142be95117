#!/bin/bash
# One gpurun call that refreshes the evidence under gpurun_out/ for profiles/: kernel-trace stats of the default bench
# command, HBM traffic of the env-step kernel (two PMC passes), SQ counters of the fused MLP rollout, of the GRU lockstep
# rollout (E = 5) and of the C3 lander rollout.  Copy / summarise into profiles/ afterwards (tools/collect_*.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/prof_bench.sh > gpurun_out/prof_bench.txt 2>&1; tail -2 gpurun_out/prof_bench.txt
bash tools/prof_pmc.sh > gpurun_out/pmc_stdout.txt 2>&1; tail -4 gpurun_out/pmc_stdout.txt
bash tools/prof_sq.sh --no-extras > gpurun_out/sq_rollout.txt 2>&1; tail -3 gpurun_out/sq_rollout.txt
mkdir -p gpurun_out/sq_mlp && rm -rf gpurun_out/sq_mlp/* && mv gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_mlp/
bash tools/prof_sq.sh --no-extras --gru > gpurun_out/sq_gru.txt 2>&1; tail -3 gpurun_out/sq_gru.txt
mkdir -p gpurun_out/sq_gru && rm -rf gpurun_out/sq_gru/* && mv gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_gru/
bash tools/prof_sq_c3.sh 4096 > gpurun_out/sq_c3.txt 2>&1; tail -3 gpurun_out/sq_c3.txt
python tools/bench_configs.py > gpurun_out/configs.jsonl 2> gpurun_out/configs.err; tail -2 gpurun_out/configs.jsonl | cut -c1-200

#!/bin/bash
# The evidence round of a build (one gpurun call, ~8 minutes): GPU tests, every counter / trace pass of tools/gpu_profile_round.sh
# plus the MFMA and Box2D MLP passes, the summaries (with the kernels' machine-code hashes) written to profiles/ AND staged
# under gpurun_out/final/ (gpurun merges only gpurun_out/ back), fuzz at depth, the env-step A/B of this box, the bench
# lines, loop timings.   usage: tools/gpu_final_round.sh [round tag = r03] [fuzz seconds per process = 240]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=${1:-r06}
FUZZ=${2:-240}
OUT=gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $OUT/${TAG}_smoke.txt
python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/${TAG}_pytest_gpu.log
bash tools/gpu_profile_round.sh > $OUT/profile_round.txt 2>&1; tail -3 $OUT/profile_round.txt | cut -c1-200
bash tools/prof_mfma.sh > gpurun_out/sq_mfma.txt 2>&1
bash tools/prof_sq_box2d.sh 4096 > gpurun_out/sqb2.txt 2>&1
python tools/collect_pmc.py $TAG
python tools/collect_sq.py $TAG rollout k_rollout_cartpole_mlp gpurun_out/sq_mlp/sq_1 gpurun_out/sq_mlp/sq_2
python tools/collect_sq.py $TAG gru_lockstep k_rollout_gru_lockstep gpurun_out/sq_gru/sq_1 gpurun_out/sq_gru/sq_2
python tools/collect_sq.py $TAG c3_lander LanderLs gpurun_out/sqc3_1 gpurun_out/sqc3_2
python tools/collect_sq.py $TAG gru_mfma k_rollout_gru_mfma gpurun_out/mf_1 gpurun_out/mf_2
# round 6: the same counters for the 4x4x1 MFMA kernel (eval_ep_num 8)
bash tools/prof_mfma.sh --eval-ep-num 8 > gpurun_out/sq_mfma4.txt 2>&1
python tools/collect_sq.py $TAG gru_mfma4 k_rollout_gru_mfma4 gpurun_out/mf_1 gpurun_out/mf_2
python tools/collect_sq.py $TAG box2d_mlp k_rollout_box2d_mlp gpurun_out/sqb2_1 gpurun_out/sqb2_2
cp profiles/${TAG}_pmc_env_step.json profiles/${TAG}_sq_*.json $OUT/
cp gpurun_out/kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
grep "^{\"metric\"" gpurun_out/prof_kt_bench.log | tail -1 > $OUT/${TAG}_bench_profiled.json
cp gpurun_out/pmc_stdout.txt $OUT/${TAG}_pmc_stdout.txt
cp gpurun_out/configs.jsonl $OUT/${TAG}_configs.jsonl
python tools/timeline_gaps.py $(find gpurun_out/prof_kt -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_generation_timeline.txt 2>&1
rm -f gpurun_out/fuzz_[0-9]*.txt
bash tools/fuzz_round.sh $FUZZ 300 > $OUT/${TAG}_fuzz_parity.txt 2>&1; tail -1 $OUT/${TAG}_fuzz_parity.txt | cut -c1-300
grep -h MISMATCH gpurun_out/fuzz_[0-9]*.txt | head -5
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I simple-es_amd/csrc tools/envstep_ab.hip -o tools/envstep_ab
tools/envstep_ab 24 15 20 200 > $OUT/${TAG}_envstep_ab_final_box.txt 2>&1; head -3 $OUT/${TAG}_envstep_ab_final_box.txt
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python bench.py --gru --no-extras --no-cpu-baseline > $OUT/${TAG}_bench_gru.json 2>> $OUT/bench.err
python bench.py --gru --eval-ep-num 16 --no-extras --no-cpu-baseline --no-roofline > $OUT/${TAG}_bench_gru16.json 2>> $OUT/bench.err
python bench.py --gru --eval-ep-num 8 --no-extras --no-cpu-baseline --no-roofline > $OUT/${TAG}_bench_gru8.json 2>> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_flags.json 2>> $OUT/bench.err
for cfg in cartpole_openai.yaml cartpole.yaml cartpole_pomdp_gru.yaml simplespread.yaml lunarlander_openai.yaml; do
  python tools/time_loop.py $cfg 2>&1 | tail -1
done > $OUT/${TAG}_time_loop.txt
python tools/time_loop.py lunarlander.yaml 0 300 2>&1 | tail -1 >> $OUT/${TAG}_time_loop.txt
python tools/time_loop.py bipedalwalker.yaml 0 60 2>&1 | tail -1 >> $OUT/${TAG}_time_loop.txt
cat $OUT/${TAG}_time_loop.txt
python tools/time_box2d_mlp.py 4096 0 2>/dev/null > $OUT/${TAG}_box2d_mlp.txt; cat $OUT/${TAG}_box2d_mlp.txt | cut -c1-160
python tools/walker_breakdown.py 4096 0 2>/dev/null > $OUT/${TAG}_walker_breakdown.txt
# phase shares of an env step: a development build with timers (never the shipped library)
mkdir -p ab
SES_OUT=$R/ab/libT.so SES_OBJ=/tmp/objT bash simple-es_amd/csrc/build.sh -DSES_PHASE_TIMERS > /dev/null 2>&1
for w in walker lander c3; do SES_LIB_PATH=$R/ab/libT.so python tools/walker_phases.py $w 4096 2>/dev/null; done > $OUT/${TAG}_step_phases.txt
python tools/c3_breakdown.py > $OUT/${TAG}_c3_by_horizon.txt 2>&1
python tools/time_small_populations.py 2>/dev/null > $OUT/${TAG}_small_populations.txt
# round 6: dependent-issue latencies of a lone wave (input of tools/chain_model.py), the MFMA forms of the GRU gate contraction, the
# tail of the strong line (4096 rows in total) by GPU count
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/dep_latency.hip -o tools/dep_latency 2>/dev/null && tools/dep_latency > $OUT/${TAG}_dep_latency.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/mfma_vs_valu_gru.hip -o tools/mfma_vs_valu_gru 2>/dev/null && tools/mfma_vs_valu_gru > $OUT/${TAG}_mfma_vs_valu_gru.txt
SES_TAIL_SHAPES=1x4096,2x2048,4x1024,8x512 python tools/time_tail.py 2>/dev/null > $OUT/${TAG}_time_tail_strong.txt; cat $OUT/${TAG}_time_tail_strong.txt
python tools/lander_step_cost.py > $OUT/${TAG}_lander_step_cost.txt 2>&1
# the multi-rank side (tests excluded: they ran above): exchange latency by kind, a generation of two ranks with and without the
# fused exchanges, the openai_es tail replicated / in shard form, the 8-rank shapes by kernel
bash tools/gpu_multirank_round.sh notests > $OUT/multirank_round.txt 2>&1
for f in time_allgather time_multirank_generation time_tail_sharded tail_by_kernel; do cp gpurun_out/${TAG}_$f.txt $OUT/ 2>/dev/null; done
# bench.py --gpus 2 / 4 rehearsed with the ranks sharing this GPU (gloo control plane, peer-store transport): the multi-rank
# code path of the bench, NOT a scaling measurement
for n in 2 4; do
  SES_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus $n --steps 100 --warmup 20 --blocks 9 --min-timed-seconds 2 --no-roofline --no-cpu-baseline --loop-generations 300 > $OUT/${TAG}_bench_rehearsal_gloo_${n}ranks_one_gpu.json 2>> $OUT/bench.err
done
# gpurun copies gpurun_out/ back only below 64 MiB: the raw rocprofv3 directories stay on the box
find gpurun_out -mindepth 1 -maxdepth 1 ! -name final -exec rm -rf {} +
du -sh gpurun_out | cut -f1
ls $OUT

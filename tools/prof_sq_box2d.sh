#!/bin/bash
# SQ counters of the Box2D MLP rollouts (LunarLanderContinuous-v2 and BipedalWalker-v3, 4096 offspring x 5 episodes x
# <= 300 steps): two passes of <= 8 SQ counters over tools/time_box2d_mlp.py, kernel-trace only.
# Summarise with: python tools/collect_sq.py r02 box2d_mlp k_rollout_box2d_mlp gpurun_out/sqb2_1 gpurun_out/sqb2_2
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/sqb2_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/sqb2_$i -- python3 $R/tools/time_box2d_mlp.py "${1:-4096}" 0 > $R/gpurun_out/sqb2_$i.log 2>&1
  tail -2 $R/gpurun_out/sqb2_$i.log | cut -c1-200
done

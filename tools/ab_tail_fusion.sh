#!/bin/bash
# A/B of the update-inside-the-perturbation-launch fusion of the openai_es tail on the bench's headline generation, interleaved
# (three rounds): "" = three launches (k_es_apply_perturb) | fused_apply_perturb=0 = four (round 4's shape before it).
# usage: tools/ab_tail_fusion.sh [out file]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
out=${1:-gpurun_out/ab_tail_fusion.txt}; : > $out
IFS='|' read -r -a variants <<< "${SES_AB_VARIANTS:-default|fused_apply_perturb=0}"     # '|'-separated SES_TUNING strings; "default" = none
for round in 1 2 3; do
  for t in "${variants[@]}"; do
    [ "$t" = default ] && t=""
    SES_TUNING="$t" python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('round $round tuning=\"$t\" ms_per_generation', round(d['ms_per_step'],4), 'env_steps_per_s %.4e' % d['value'])" >> $out
  done
done
cat $out

#!/bin/bash
# rocprofv3 kernel trace of the default bench command; summaries land in gpurun_out/prof_*
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/prof_kt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_kt_bench.log 2>&1
grep "^{\"metric\"" $R/gpurun_out/prof_kt_bench.log | tail -1 | cut -c1-300
find $R/gpurun_out/prof_kt -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$R'/gpurun_out/kernel_stats.csv; head -20 {}'

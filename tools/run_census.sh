#!/bin/bash
# development helper: dispatcher census + env-step stream-skew experiment
mkdir -p gpurun_out
: > gpurun_out/census.log
for g in "640 64" "1280 64" "2560 64" "320 256" "256 320" "160 256" "1024 64" "512 128"; do
  ./tools/census $g >> gpurun_out/census.log 2>&1
done
./tools/envstep_tune 24 2>&1 | grep skew > gpurun_out/skew24.log
cat gpurun_out/census.log gpurun_out/skew24.log

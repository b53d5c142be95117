#!/bin/bash
# Builds libses_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [extra hipcc flags]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${SES_OUT:-$HERE/../libses_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -fno-slp-vectorize: v_pk_fma_f32 has no throughput advantage on gfx950 (a packed op costs two issue slots)
# and SLP packing adds v_mov traffic; measured 0.357 -> 0.311 ms on the 4096x5x500 rollout.
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize
       -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I/opt/rocm/include)
"$HIPCC" "${FLAGS[@]}" "$@" "$HERE/ses_core.hip" "$HERE/ses_rollout.hip" "$HERE/ses_strategy.hip" "$HERE/ses_comm.hip" -ldl -o "$OUT"
echo "built $OUT"

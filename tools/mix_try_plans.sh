#!/bin/bash
# Run on the GPU box: per-pass times of alternative plans of one length.  bash tools/mix_try_plans.sh "nil nxl" "N:plan" "N:plan" ...
SHAPE=$1; shift
for F in "$@"; do
  P3D_GEN_FORCE="$F" python tools/gen_mix_plans.py --min 120 --max 4096 --parts 8 > pseudo-3d-interpolation_amd/csrc/p3d_mix_plans.inc
  make -C pseudo-3d-interpolation_amd/csrc -j16 > /dev/null 2>&1 || { echo "build failed for $F"; continue; }
  echo -n "$F -> "; bash tools/mix_passes.sh "$SHAPE"
done

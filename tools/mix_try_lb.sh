#!/bin/bash
# Run on the GPU box: rows per workgroup of the row pass of one plan.  bash tools/mix_try_lb.sh "nil nxl" N "plan" colt lb lb lb ...
SHAPE=$1; N=$2; PLAN=$3; COLT=$4; shift 4
cp pseudo-3d-interpolation_amd/csrc/p3d_mix_plans.inc /tmp/p3d_mix_plans.inc.keep
for LB in "$@"; do
  P3D_GEN_FORCE="$N:$PLAN" P3D_GEN_FORCE_COLT=$COLT P3D_GEN_FORCE_LB=$LB python tools/gen_mix_plans.py --only $N --parts 8 > pseudo-3d-interpolation_amd/csrc/p3d_mix_plans.inc
  make -C pseudo-3d-interpolation_amd/csrc -j16 > /tmp/mk.log 2>&1 || { echo "build failed for LB=$LB"; grep -m3 error /tmp/mk.log; continue; }
  echo -n "N=$N plan $PLAN rows/wg $LB -> "; bash tools/mix_passes.sh "$SHAPE"
done
cp /tmp/p3d_mix_plans.inc.keep pseudo-3d-interpolation_amd/csrc/p3d_mix_plans.inc

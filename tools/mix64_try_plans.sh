#!/bin/bash
# Run on the GPU box: the double-precision loop with alternative plans of one length.  bash tools/mix64_try_plans.sh "N:plan[@colt[@lb]]" ...
for F in "$@"; do
  IFS=@ read PLAN COLT LB <<< "$F"
  N=${PLAN%%:*}   # (only that length is instantiated: a minute of build time less per variant; NIL / NXL of tools/f64_bench.py should be N)
  P3D_GEN_FORCE="$PLAN" P3D_GEN_FORCE_COLT=${COLT:-0} P3D_GEN_FORCE_LB=${LB:-0} python tools/gen_mix_plans.py --f64 --only $N --parts 8 > pseudo-3d-interpolation_amd/csrc/p3d_mix64_plans.inc
  make -C pseudo-3d-interpolation_amd/csrc -j16 > /dev/null 2>&1 || { echo "build failed for $F"; continue; }
  echo -n "$F -> "; python tools/f64_bench.py | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_iteration'],4), 'ms/it', round(d['Gpt_per_s'],2), 'Gpt/s')"
done

#!/bin/bash
# Run on the GPU box: the double-precision FFT loop on square-ish shapes whose extents have a plan on the register engine, against the LDS-image passes
# (P3D_NO_MIX64=1).   bash tools/f64_sweep.sh "625 700" "840 840" ...
for shp in "$@"; do set -- $shp
  NS=$(( 33554432 / ($1 * $2) )); [ $NS -lt 2 ] && NS=2; [ $NS -gt 64 ] && NS=64
  a=$(NIL=$1 NXL=$2 NS=$NS NCHECK=0 python tools/f64_bench.py | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['Gpt_per_s'],1))")
  b=$(P3D_NO_MIX64=1 NIL=$1 NXL=$2 NS=$NS NCHECK=0 python tools/f64_bench.py | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['Gpt_per_s'],1))")
  echo "$1 x $2 x $NS: register engine $a Gpt/s, LDS-image passes $b Gpt/s"
done

#!/bin/bash
# compare experimental library builds on the GPU box: [NITER="10 100"] bash tools/variants.sh libA.so[:ENV=1] libB.so ...
for niter in ${NITER:-10}; do
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=${spec#*:}
  for rep in 1 2; do
    echo -n "niter $niter $spec: "; env $envs P3D_LIB_PATH=$PWD/pseudo-3d-interpolation_amd/$lib timeout -k 10 120 python3 tools/pocs_driver.py --niter $niter | tail -1
  done
done
done

#!/bin/bash
# compare experimental library builds on the GPU box: bash tools/variants.sh libA.so[:ENV=1] libB.so ...
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=${spec#*:}
  for rep in 1 2; do
    echo -n "$spec: "; env $envs P3D_LIB_PATH=$PWD/pseudo-3d-interpolation_amd/$lib timeout -k 10 120 python3 tools/pocs_driver.py --niter 10 | tail -1
  done
done

#!/bin/bash
# ablations of the mix kernels: per-pass ms (pocs_driver prints plan.last_profile())
for abl in 0 1 2 4 8 9 15; do
  echo -n "abl $abl: "; P3D_MIX_ABL=$abl timeout -k 10 120 python tools/pocs_driver.py --nil ${1:-1000} --nxl ${2:-1000} --nslices 128 --niter 20 2>&1 | tail -1
done
echo -n "flex: "; P3D_NO_MIX=1 timeout -k 10 120 python tools/pocs_driver.py --nil ${1:-1000} --nxl ${2:-1000} --nslices 128 --niter 20 2>&1 | tail -1
echo -n "tuned 1024: "; timeout -k 10 120 python tools/pocs_driver.py --nil 1024 --nxl 1024 --nslices 128 --niter 20 2>&1 | tail -1

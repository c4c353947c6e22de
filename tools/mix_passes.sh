#!/bin/bash
# per-pass ms of the FFT POCS loop for a list of "nil nxl" shapes (128 slices, 20 iterations; tools/pocs_driver.py prints plan.last_profile())
for shp in "$@"; do
  set -- $shp
  echo -n "$1 x $2: "; timeout -k 10 120 python tools/pocs_driver.py --nil $1 --nxl $2 --nslices 128 --niter 20 2>&1 | tail -1 | sed -e "s/'colpass_launches': [0-9]*, //" -e "s/, 'rowpass_launches': [0-9]*//"
done

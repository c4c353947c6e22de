#!/bin/bash
# HBM traffic of the dense path (P3D_NO_SPARSE=1): the two traffic counters only.  bash tools/profile_dense.sh <tag>
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${TAG}_dense
mkdir -p $OUT
export TMPDIR=/tmp
export P3D_NO_SPARSE=1
cd /tmp
for PASS in FETCH_SIZE WRITE_SIZE; do
  python3 $REPO/tools/pmc_slots.py $PASS > /dev/null || { echo "counter list beyond one pass: $PASS"; exit 2; }
  timeout -k 10 300 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT/pmc_$PASS -- python3 $REPO/tools/pocs_driver.py --niter ${PMC_NITER:-100} > $OUT/pmc_$PASS.log 2>&1 || { echo "pmc pass failed: $PASS"; tail -5 $OUT/pmc_$PASS.log; }
done
cd $REPO
python3 tools/summarize_prof.py $OUT gpurun_out/profiles_${TAG}_dense

#!/bin/bash
# Run on the GPU box: SQ counter passes over tools/pocs_driver.py for one shape -> gpurun_out/stall_<tag>/
# usage: bash tools/stall_probe.sh <nil> <nxl> <nslices> <tag>
set -o pipefail
NIL=${1:-1000}; NXL=${2:-1000}; NS=${3:-128}; TAG=${4:-probe}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/stall_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
DRV="$REPO/tools/pocs_driver.py --nil $NIL --nxl $NXL --nslices $NS --niter 6"
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 $DRV > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:70]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if not any(s in k for s in ("row", "col")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        v = sorted(v)
        print(f"   {c:24s} median {v[len(v)//2]:.4g}  n={len(v)}")
PY

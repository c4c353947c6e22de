#!/bin/bash
# Run on the GPU box: SQ counter passes over bench.py --config <c> --only-main (as tools/stall_probe.sh, for the configurations that pocs_driver.py does not cover)
# usage: bash tools/stall_probe_cfg.sh <config> <steps>
set -o pipefail
CFG=${1:-3}; STEPS=${2:-20}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/stall_cfg$CFG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="$REPO/bench.py --config $CFG --only-main --no-cpu-baseline --no-dense --repeats 1 --warmup 0 --steps $STEPS"
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 $BENCH > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:90]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = len(cs.get("SQ_WAVES", []))
    if n < 3:
        continue
    med = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}
    busy = med.get("SQ_BUSY_CYCLES", 0) / 32.0          # cycles of the dispatch (32 shader engines count)
    valu = med.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * busy) if busy else 0
    lds = med.get("SQ_ACTIVE_INST_LDS", 0) * 4 / (1024 * busy) if busy else 0
    print(f"{k}\n   dispatches {n}, cycles {busy:.3g}, vector ALU busy {valu:.2f}, LDS instructions active {lds:.2f} of the SIMDs' cycles; VALU wave instructions {med.get('SQ_INSTS_VALU', 0):.3g}")
PY

#!/bin/bash
# Run on the GPU box:  bash tools/pmc_kernels.sh <outdir-name> "<counters>" <script.py> [args]  -> per-kernel counter means (torch-free scripts only)
# The counter list goes through tools/pmc_slots.py first: a list beyond one pass's slots (e.g. FETCH_SIZE + WRITE_SIZE = 5 of 4 TCC
# slots) is SPLIT into passes that fit, run one after the other -- rocprofv3 does not refuse such a request cleanly on this image
# (profiles/r03_pmc_counter_budget.txt).  A pass that fails stops the script: no further GPU step after a failure.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=$1; PMC="$2"; shift 2
export TMPDIR=/tmp
mapfile -t PASSES < <(python3 $R/tools/pmc_slots.py --split $PMC) || exit 2
[ ${#PASSES[@]} -gt 0 ] || { echo "no counters"; exit 2; }
i=0
for PASS in "${PASSES[@]}"; do
  OUT=$R/gpurun_out/${NAME}_p$i; i=$((i+1))
  python3 $R/tools/pmc_slots.py $PASS > /dev/null || exit 2          # belt and braces: the pass itself must fit
  cd /tmp
  timeout -k 10 150 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT -- python3 $R/$1 "${@:2}" > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
  cd $R
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  echo "== pass: $PASS"
  python3 - "$f" <<PY
import csv,collections,sys
agg=collections.defaultdict(lambda:collections.defaultdict(lambda:[0.0,0]))
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]; n=n.split("::")[-1][:28] if "::" in n else n[:28]
    k=(n, r["Grid_Size"])
    a=agg[k][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k,cs in sorted(agg.items(), key=lambda kv:-sum(v[1] for v in kv[1].values()))[:8]:
    print(k, {c:"%.4g"%(v[0]/v[1]) for c,v in cs.items()})
PY
done

#!/bin/bash
# Run on the GPU box:  bash tools/pmc_kernels.sh <outdir-name> "<counters>" <script.py>   -> per-kernel counter means (torch-free scripts only)
# ONE TCC counter (FETCH_SIZE, WRITE_SIZE ...) per call: two of them in one pass hung the run on this pool
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$1; PMC="$2"; shift 2
export TMPDIR=/tmp; cd /tmp
timeout -k 10 150 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT -- python3 $R/$1 > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
cd $R
f=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$f" <<PY
import csv,collections,sys
agg=collections.defaultdict(lambda:collections.defaultdict(lambda:[0.0,0]))
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]; n=n.split("::")[-1][:28] if "::" in n else n[:28]
    k=(n, r["Grid_Size"])
    a=agg[k][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k,cs in sorted(agg.items(), key=lambda kv:-sum(v[1] for v in kv[1].values()))[:6]:
    print(k, {c:"%.4g"%(v[0]/v[1]) for c,v in cs.items()})
PY

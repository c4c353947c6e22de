#!/bin/bash
# Run on the GPU box:  bash tools/prof_kernels.sh <outdir-name> <script.py> [env...]   -> per-kernel / per-grid time table
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$1; shift
export TMPDIR=/tmp; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/$1 > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
cd $R
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<PY
import csv,collections,sys
tr=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda:[0,0])
for r in tr:
    n=r["Kernel_Name"]; n=n.split("::")[-1][:44] if "::" in n else n[:44]
    k=(n, r["Grid_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"])
    agg[k][0]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); agg[k][1]+=1
tot=sum(v[0] for v in agg.values())
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][0])[:16]:
    print("%-46s grid %8s lds %6s vgpr %3s  n %5d  %9.1f us avg  %5.1f%%" % (k[0],k[1],k[2],k[3],v[1],v[0]/v[1]/1e3,100*v[0]/tot))
PY

#!/bin/bash
# Run on the GPU box:  bash tools/wv_trace.sh <tag>   -> per-launch-shape durations of the wavelet tile kernels (tools/wavelet_bench.py)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$1
export TMPDIR=/tmp K=${K:-10} NCHECK=0; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/wavelet_bench.py > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
cd $R
python3 - "$(find $OUT -name '*kernel_trace.csv' | head -1)" <<PY
import csv, collections, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    n = ("idwt2" if "idwt2_tile" in n else "dwt2" if "dwt2_tile" in n else n.split("(")[0][-30:])
    agg[(n, int(r.get("Grid_Size", r.get("Grid_Size_X", 0))))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v); print(k, "calls", len(v), "median_us %.1f" % (v[len(v) // 2] / 1e3), "min_us %.1f" % (v[0] / 1e3))
PY

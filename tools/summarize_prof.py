"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into small tracked summaries."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def short(name):
    m = re.search(r"(row_kernel|col_kernel)<(\d+)(?:, *(\d+))?, *(\d+)>", name)
    if m:
        kind, n, t, mode = m.group(1), m.group(2), m.group(3), m.group(4)
        if kind == "row_kernel":
            return f"row_kernel<{n},{['FIRST', 'MID', 'LAST'][int(t if m.group(3) and not mode else mode)]}>" if False else f"row_kernel<{n},mode{mode}>"
        return f"col_kernel<{n},T{t},mode{mode}>"
    return name.split("(")[0][:60]


# ---- kernel trace ------------------------------------------------------------------------------
rows = []
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = defaultdict(list)
meta = {}
for r in rows:
    k = short(r["Kernel_Name"])
    agg[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    meta[k] = (r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""),
               r.get("LDS_Block_Size", ""), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("Grid_Size", r.get("Grid_Size_X", "")))
total = sum(sum(v) for v in agg.values()) or 1
with open(os.path.join(dst, "kernel_stats.csv"), "w") as out:
    out.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent,vgpr,agpr,sgpr,lds_bytes,wg_size,grid\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        m = meta[k]
        out.write(f"\"{k}\",{len(v)},{sum(v) / 1e6:.3f},{sum(v) / len(v) / 1e3:.1f},{min(v) / 1e3:.1f},{max(v) / 1e3:.1f},"
                  f"{100.0 * sum(v) / total:.1f},{m[0]},{m[1]},{m[2]},{m[3]},{m[4]},{m[5]}\n")
print(open(os.path.join(dst, "kernel_stats.csv")).read())

# ---- PMC passes ----------------------------------------------------------------------------------
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, cs in pmc.items():
    summary[k] = {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()}
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
for k, cs in sorted(summary.items()):
    line = ", ".join(f"{c}={d['mean']:.4g}" for c, d in sorted(cs.items()))
    print(f"{k}: {line}")

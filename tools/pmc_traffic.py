"""HBM bytes per POCS iteration from the PMC summary of tools/profile.sh: (FETCH_SIZE x 2 + WRITE_SIZE) KiB summed over the two kernels
of an iteration (FETCH_SIZE counts 64-byte requests where the L2 issues 128-byte ones on gfx950: MI355X_MICROARCH.md, HBM)."""
import json
import os
import re
import sys

src, niter = sys.argv[1], int(sys.argv[2])
pmc = json.load(open(os.path.join(src, "pmc_summary.json")))
line = json.loads(open(os.path.join(src, "bench_line_under_trace.json")).read())
parts = {}
for k, cs in pmc.items():
    steady_row = re.search(r"row_pipe64_kernel<\d+, \d, (true|false), 0,", k) is not None   # PM = PIPE_MID only (first / last pass: once per job)
    if ("col_kernel" in k and k.endswith("mode0>")) or steady_row or "row_pipe_kernel" in k:
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            parts[k] = {"read_bytes": 2 * cs["FETCH_SIZE"]["mean"] * 1024, "written_bytes": cs["WRITE_SIZE"]["mean"] * 1024,
                        "launches": cs["FETCH_SIZE"]["launches"], "tcc_miss_x128": cs.get("TCC_MISS_sum", {}).get("mean", 0) * 128}
total = sum(p["read_bytes"] + p["written_bytes"] for p in parts.values())
wl = line["config"]["workload"].split(" ")[0]
print(json.dumps({
    "workload": wl, "iterations": niter, "hbm_bytes_per_iteration": int(total), "kernels": parts,
    "nonzero_block_fraction": line.get("sparse_spectrum", {}).get("nonzero_block_fraction"),
    "method": "rocprofv3 --pmc passes of tools/profile.sh on bench.py itself (--no-cpu-baseline --no-dense --repeats 1 --warmup 0): per-launch "
              "means over the col_kernel<COL_ITER> and persistent-row-pass launches of the trace; (FETCH_SIZE x 2 [gfx950 wide-read "
              "correction, MI355X_MICROARCH.md HBM] + WRITE_SIZE) x 1024 B; cross-check column tcc_miss_x128 = TCC_MISS_sum x 128 B",
    "round": 2}, indent=1))

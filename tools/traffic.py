#!/usr/bin/env python3
"""profiles/traffic_config<i>.json from the rocprofv3 passes of tools/traffic.sh (kernel trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE over
`bench.py --config i --only-main ...`): HBM bytes per POCS iteration of the loop's kernels,

    bytes = (FETCH_SIZE x 2 + WRITE_SIZE) x 1024      (FETCH_SIZE counts 64-byte requests where the L2 of gfx950 issues 128-byte ones:
                                                       /opt/skills/guides/MI355X_MICROARCH.md, HBM; WRITE_SIZE is exact for streaming stores)

summed over the dispatches of the loop's kernels and divided by the iterations the trace holds (= dispatches of the kernel that runs once
per iteration).  The file is stamped with the hash of the kernel sources it was measured on: bench.py reports `roofline.traffic` only
while that hash matches."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash, CONFIGS)

src, config, steps, rnd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = bench.CONFIGS[config]
kind = cfg["kind"]
# (pattern of the loop's kernels, pattern of the one that runs once per iteration and rank)
LOOP = {
    "FFT": (r"col_kernel<\d+, \d+, (0|4|5)>|col_pipe_kernel|row_pipe64_kernel<\d+, \d, (true|false), 0,|row_pipe32_kernel<\d, (true|false), 0,|row_pipe_kernel|row_real_kernel|resident_kernel",
            r"col_kernel<\d+, \d+, (0|4|5)>|col_pipe_kernel|resident_kernel"),
    "WAVELET": (r"dwt2_tile_kernel|idwt2_tile_kernel|wcoarse_kernel|wfuse1_kernel", r"wcoarse_kernel"),
    "SHEARLET": (r"p3d::row_kernel|p3d::col_kernel|col_pipe_kernel|col_shear_pair_kernel|supdate_kernel|mirror_rows_kernel", r"col_shear_pair_kernel|col_pipe_kernel"),
}[kind]


def counters(name):
    out = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(src, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"]
            out[k][0] += float(r["Counter_Value"])
            out[k][1] += 1
    return out


fetch, write, valu = counters("FETCH_SIZE"), counters("WRITE_SIZE"), counters("SQ_INSTS_VALU")
# average duration of every kernel in the run WITHOUT counters (the kernel trace of the same command)
dur_ns = {}
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur_ns[r["Name"]] = float(r["AverageNs"])
# one wavefront instruction per SIMD every two cycles (64 lanes on a SIMD-32): 256 CUs x 4 SIMDs x 2.4 GHz / 2 -- the rate behind the
# 157.3 TFLOP/s FP32 vector peak of /opt/skills/guides/MI355X_MICROARCH.md (x 64 lanes x 2 flop per FMA)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2
line = json.loads([l for l in open(os.path.join(src, "trace.log")) if l.startswith("{")][-1])
loop_re, once_re = re.compile(LOOP[0]), re.compile(LOOP[1])
kernels, read_b, written_b, iters = {}, 0.0, 0.0, 0
for k in sorted(set(fetch) | set(write)):
    if not loop_re.search(k):
        continue
    rb, wb = 2 * fetch[k][0] * 1024, write[k][0] * 1024
    n = max(fetch[k][1], write[k][1])
    short = re.sub(r"\(.*$", "", k.replace("(anonymous namespace)::", "").replace("void ", ""))[:90]
    if short in kernels:
        short = short + f" #{len(kernels)}"
    kernels[short] = {"dispatches": n, "read_bytes_per_dispatch": rb / max(fetch[k][1], 1), "written_bytes_per_dispatch": wb / max(write[k][1], 1)}
    if k in valu and valu[k][1] and k in dur_ns:
        per = valu[k][0] / valu[k][1]
        kernels[short].update({"valu_wave_instructions_per_dispatch": per, "average_ns": dur_ns[k],
                               "valu_issue_frac": per / (dur_ns[k] * 1e-9) / VALU_ISSUE_PEAK})
    read_b += rb
    written_b += wb
    if once_re.search(k):
        iters += fetch[k][1]
if kind == "FFT" and any("resident_kernel" in k for k in kernels):
    iters *= steps                                   # one dispatch = one whole job of `steps` iterations
n_local = line["config"]["slices_per_gpu"]
per_it = (read_b + written_b) / max(iters, 1)
rec = {
    "workload": line["config"]["workload"],
    "workload_key": (f"{cfg['nil']}x{cfg['nxl']}x{n_local}" if kind != "SHEARLET" else f"{cfg['nil']}x{cfg['nxl']}x{125}sh"),
    "iterations": steps, "iterations_in_trace": iters,
    "hbm_bytes_per_iteration": per_it / (n_local if kind == "SHEARLET" else 1),
    "per": "slice-iteration" if kind == "SHEARLET" else "iteration of the whole cube",
    "read_bytes_per_iteration": read_b / max(iters, 1), "written_bytes_per_iteration": written_b / max(iters, 1),
    "kernels": kernels,
    "nonzero_block_fraction": (line.get("sparse_spectrum") or {}).get("nonzero_block_fraction"),
    "method": "tools/traffic.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_slots.py) over bench.py --config "
              f"{config} --only-main --no-cpu-baseline --no-dense --repeats 1 --warmup 0 --steps {steps}; (FETCH_SIZE x 2 [gfx950 wide-read correction] + "
              "WRITE_SIZE) x 1024 B summed over the loop's kernels / iterations in the trace",
    "valu_issue_peak_wave_instructions_per_s": VALU_ISSUE_PEAK,
    "valu_note": "SQ_INSTS_VALU counts wavefront instructions; a packed-f32 instruction (v_pk_*: two results per lane) counts once",
    "kernel_source_hash": bench.kernel_source_hash(), "round": rnd,
}
print(json.dumps(rec, indent=1))

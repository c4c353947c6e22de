#!/bin/bash
# Run on the GPU box (via gpurun): the evidence of round 5 that is kept under profiles/ -- part A (traffic + kernel statistics of the four
# measured configurations) or part B (7-smooth grids, the double-precision loops).   bash tools/collect_r05.sh A|B
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profiles_r05
mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
if [ "$1" = "A" ]; then
  for CFG in 2 1 3 4; do
    STEPS=100; [ $CFG = 1 ] && STEPS=50; [ $CFG = 3 ] && STEPS=50
    bash tools/traffic.sh $CFG $STEPS 5 > $OUT/traffic_c$CFG.log 2>&1 || { echo "traffic config $CFG failed"; tail -5 $OUT/traffic_c$CFG.log; exit 1; }
    cp gpurun_out/traffic_config$CFG.json $OUT/; cp gpurun_out/kernel_stats_config$CFG.csv $OUT/r05_kernel_stats_config$CFG.csv
    echo "config $CFG done"
  done
else
  echo "# FFT POCS loop on 7-smooth extents (mixed-radix register engine, p3d_mix.hpp), 128 slices x 20 iterations, tools/shape_sweep.sh" > $OUT/r05_mix_shape_sweep.txt
  bash tools/shape_sweep.sh "1000 1000" "2000 1500" "1200 1600" "960 768" "900 900" "600 500" "1500 2000" "2400 1800" "1024 1024" >> $OUT/r05_mix_shape_sweep.txt 2>&1 || exit 1
  echo "# the same shapes on the LDS-image passes of p3d_flex.hip (P3D_NO_MIX=1)" >> $OUT/r05_mix_shape_sweep.txt
  P3D_NO_MIX=1 bash tools/shape_sweep.sh "1000 1000" "2000 1500" "1200 1600" "960 768" >> $OUT/r05_mix_shape_sweep.txt 2>&1 || exit 1
  echo "# per-pass ms (column pass / row pass; tools/pocs_driver.py, 128 slices)" >> $OUT/r05_mix_shape_sweep.txt
  bash tools/mix_passes.sh "1000 1000" "2000 1500" "1200 1600" "960 768" "900 900" >> $OUT/r05_mix_shape_sweep.txt 2>&1
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/smooth_trace -- python3 $REPO/bench.py --nil 1000 --nxl 1000 --nslices 128 --steps 20 --only-main --no-cpu-baseline --no-dense --repeats 3 > $OUT/r05_bench_line_1000x1000.json 2> $OUT/smooth_trace.err || { tail -3 $OUT/smooth_trace.err; exit 1; }
  cp $(find $OUT/smooth_trace -name "*kernel_stats.csv" | head -1) $OUT/r05_kernel_stats_1000x1000.csv
  for PASS in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT/smooth_pmc_$PASS -- python3 $REPO/tools/pocs_driver.py --nil 1000 --nxl 1000 --nslices 128 --niter 20 > $OUT/smooth_pmc_$PASS.log 2>&1 || { echo "pmc $PASS failed"; exit 1; }
  done
  python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/smooth_pmc_{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "mix_" in r["Kernel_Name"]:
                k = "mix_row_kernel<1000>" if "mix_row_kernel" in r["Kernel_Name"] else "mix_col_kernel<1000>"
                a = agg[k][name]; a[0] += float(r["Counter_Value"]); a[1] += 1
rec = {k: {"read_bytes_per_dispatch": 2 * v["FETCH_SIZE"][0] / max(v["FETCH_SIZE"][1], 1) * 1024, "written_bytes_per_dispatch": v["WRITE_SIZE"][0] / max(v["WRITE_SIZE"][1], 1) * 1024,
           "dispatches": v["FETCH_SIZE"][1]} for k, v in agg.items()}
json.dump({"workload": "1000 x 1000 x 128 complex64, 80 % missing, hard, 20 iterations (tools/pocs_driver.py)", "method": "(FETCH_SIZE x 2 [gfx950] + WRITE_SIZE) x 1024 B per dispatch, separate --pmc passes",
           "kernels": rec}, open(f"{out}/r05_traffic_1000x1000.json", "w"), indent=1)
print(json.dumps(rec))
PY
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f64_trace -- python3 $REPO/tools/f64_bench.py > $OUT/r05_f64_line.json 2> $OUT/f64_trace.err || exit 1
  cp $(find $OUT/f64_trace -name "*kernel_stats.csv" | head -1) $OUT/r05_kernel_stats_f64_1024.csv
  cd $REPO
  for shp in "1024 1024 32" "1000 1000 32" "512 512 64" "2048 2048 8" "2000 1500 8" "960 768 32"; do set -- $shp
    echo -n "$1 x $2 x $3: register engine "; NIL=$1 NXL=$2 NS=$3 python tools/f64_bench.py | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['Gpt_per_s'],1), 'Gpt/s', round(d['ms_per_iteration'],3), 'ms per iteration', d['rel_l2_vs_oracle'])"
    echo -n "      LDS-image passes (P3D_NO_MIX64=1) "; P3D_NO_MIX64=1 NIL=$1 NXL=$2 NS=$3 python tools/f64_bench.py | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['Gpt_per_s'],1), 'Gpt/s')"
  done > $OUT/r05_f64_register_engine.txt 2>&1
  python tools/wavelet_bench.py > $OUT/r05_wavelet_bench.txt 2>&1 || true
  rm -rf $OUT/smooth_trace $OUT/smooth_pmc_FETCH_SIZE $OUT/smooth_pmc_WRITE_SIZE $OUT/f64_trace
fi
ls $OUT

#!/bin/bash
# Run on the GPU box: the bench lines and the kernel statistics of the default command that profiles/ keeps for round 5.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/final_r05
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
python bench.py --steps 20 --warmup 5 > $OUT/r05_bench_line_steps20.json 2> $OUT/steps20.err || { tail -5 $OUT/steps20.err; exit 1; }
echo "steps20 done"
python bench.py > $OUT/r05_bench_line.json 2> $OUT/default.err || { tail -5 $OUT/default.err; exit 1; }
echo "default done"
for c in 1 3 4; do python bench.py --config $c > $OUT/r05_bench_line_config$c.json 2> $OUT/config$c.err || { tail -5 $OUT/config$c.err; exit 1; }; echo "config $c done"; done
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 20 --warmup 5 --only-main --no-cpu-baseline --no-dense > $OUT/traced_line.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/r05_bench_kernel_stats.csv
rm -rf $OUT/trace
head -4 $OUT/r05_bench_kernel_stats.csv | cut -c1-200

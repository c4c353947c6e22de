#!/bin/bash
# Run on the GPU box (via gpurun):  bash tools/traffic.sh <config> <steps> [round]
# Kernel trace + one PMC pass per TCC counter (tools/pmc_slots.py: FETCH_SIZE and WRITE_SIZE do not fit one pass) of bench.py ITSELF on
# configuration <config>; writes gpurun_out/traffic_config<config>.json (copy to profiles/) and the kernel statistics beside it.
set -o pipefail
CFG=${1:-2}; STEPS=${2:-100}; RND=${3:-5}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/traffic_c$CFG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="$REPO/bench.py --config $CFG --only-main --no-cpu-baseline --no-dense --repeats 1 --warmup 0 --steps $STEPS"
echo "== kernel trace + stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $BENCH > $OUT/trace.log 2>&1 || { echo "trace run failed"; tail -5 $OUT/trace.log; exit 1; }
# (SQ_INSTS_VALU: vector instructions per dispatch -- the VALU-issue roofline of the kernels that are bound by it, tools/traffic.py)
for PASS in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  python3 $REPO/tools/pmc_slots.py $PASS > /dev/null || exit 2
  echo "== pmc $PASS"
  timeout -k 10 400 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT/pmc_$PASS -- python3 $BENCH > $OUT/pmc_$PASS.log 2>&1 || { echo "pmc pass failed: $PASS"; tail -5 $OUT/pmc_$PASS.log; exit 1; }
done
cd $REPO
python3 tools/traffic.py $OUT $CFG $STEPS $RND > gpurun_out/traffic_config$CFG.json && head -c 1500 gpurun_out/traffic_config$CFG.json
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/kernel_stats_config$CFG.csv; echo; head -8 $f | cut -c1-160

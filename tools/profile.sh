#!/bin/bash
# Run on the GPU box (via gpurun):  bash tools/profile.sh <tag> [bench args...]
# rocprofv3 kernel trace and PMC passes of bench.py ITSELF (its inputs are generated without torch RNG kernels, so counter
# collection no longer aborts there).  Raw output under gpurun_out/prof_<tag>/, summaries under gpurun_out/profiles_<tag>/ .
set -o pipefail
TAG=${1:-r02}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT $REPO/gpurun_out/profiles_$TAG
export TMPDIR=/tmp
cd /tmp
# the default bench job without its side runs and with ONE timed repeat: the trace then holds the K-iteration schedule twice (timed
# job + the profiled repeat), and every col_kernel / row_pipe64_kernel launch belongs to it
BENCH="$REPO/bench.py --no-cpu-baseline --no-dense --repeats 1 --warmup 0 --steps ${PMC_NITER:-100} $@"
echo "== kernel trace + stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $BENCH > $OUT/trace.log 2>&1 || { echo "trace run failed"; tail -5 $OUT/trace.log; exit 1; }
grep "^{" $OUT/trace.log | tail -1 > $REPO/gpurun_out/profiles_$TAG/bench_line_under_trace.json
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  python3 $REPO/tools/pmc_slots.py $PASS > /dev/null || { echo "counter list beyond one pass: $PASS"; exit 2; }
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  echo "== pmc $PASS"
  timeout -k 10 300 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT/pmc_$NAME -- python3 $BENCH > $OUT/pmc_$NAME.log 2>&1 || { echo "pmc pass failed: $PASS"; tail -5 $OUT/pmc_$NAME.log; }
done
cd $REPO
python3 tools/summarize_prof.py $OUT gpurun_out/profiles_$TAG
python3 tools/pmc_traffic.py gpurun_out/profiles_$TAG ${PMC_NITER:-100} > gpurun_out/profiles_$TAG/pmc_traffic.json && cat gpurun_out/profiles_$TAG/pmc_traffic.json

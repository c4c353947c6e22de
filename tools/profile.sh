#!/bin/bash
# Run on the GPU box (via gpurun):  bash tools/profile.sh <tag> [bench args...]
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ and summaries under gpurun_out/profiles_<tag>/ .
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT $REPO/gpurun_out/profiles_$TAG
export TMPDIR=/tmp
cd /tmp
# the default bench job (100 iterations) without its side runs: every col_kernel / row_pipe64_kernel launch in the trace belongs to
# the 100-iteration schedule, so the per-kernel averages are the colpass_ms / rowpass_ms of the bench line
BENCH="$REPO/bench.py --no-cpu-baseline --no-dense $@"
echo "== kernel trace + stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $BENCH --steps 100 --warmup 0 > $OUT/trace.log 2>&1 || { echo "trace run failed"; tail -5 $OUT/trace.log; exit 1; }
tail -1 $OUT/trace.log
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  echo "== pmc $PASS"
  timeout -k 10 300 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT/pmc_$NAME -- python3 $REPO/tools/pocs_driver.py --niter ${PMC_NITER:-100} > $OUT/pmc_$NAME.log 2>&1 || { echo "pmc pass failed: $PASS"; tail -5 $OUT/pmc_$NAME.log; }
done
cd $REPO
python3 tools/summarize_prof.py $OUT gpurun_out/profiles_$TAG

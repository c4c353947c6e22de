#!/bin/bash
# extra stall-attribution PMC passes for the POCS kernels (torch-free driver); bash tools/pmc_extra.sh <tag>
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT $REPO/gpurun_out/profiles_$TAG
export TMPDIR=/tmp
cd /tmp
i=0
for PASS in "SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  python3 $REPO/tools/pmc_slots.py $PASS > /dev/null || { echo "counter list beyond one pass: $PASS"; exit 2; }
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT/pmc_x$i -- python3 $REPO/tools/pocs_driver.py --niter 3 > $OUT/pmc_x$i.log 2>&1 || { echo "pmc pass failed: $PASS"; tail -3 $OUT/pmc_x$i.log; }
done
cd $REPO
python3 tools/summarize_prof.py $OUT gpurun_out/profiles_$TAG | grep -E "pipe|col_kernel<1024,T8,mode0>"

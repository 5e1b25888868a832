#!/bin/bash
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters.txt 2>&1
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$TAG -- python3 $GRAFT_REPO_ROOT/tools/conv_one.py > $GRAFT_REPO_ROOT/gpurun_out/pmc/$TAG.log 2>&1
done
cd $GRAFT_REPO_ROOT
find gpurun_out/pmc -name "*counter_collection.csv" | head

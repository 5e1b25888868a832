#!/bin/bash
# MFMA utilisation / LDS / wait counters of the bench step, per kernel variant (north_star: "MFMA utilisation vs gfx950 peak").
# Two --pmc passes (8 SQ slots each; GRBM_GUI_ACTIVE rides in both as the cycle base), kernel trace in the same runs, the
# program directly after `--` (no env / bash -c hop).
#   bash tools/gpu_pmc_mfma.sh [TAG [bench.py args...]]   -> gpurun_out/pmcm_TAG/summary.json (copy to profiles/rNN_TAG_pmc_mfma.json)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT
TAG=${1:-r18}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcm_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmcm_A /tmp/pmcm_B
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/pmcm_A -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 4 --no_cpu_baseline --no_roofline --no_graph --no_literal "$@" > $OUT/passA.log 2>&1
timeout 900 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/pmcm_B -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 4 --no_cpu_baseline --no_roofline --no_graph --no_literal "$@" > $OUT/passB.log 2>&1
cd $GRAFT_REPO_ROOT
grep -h "Unable to find\|Missing" $OUT/passA.log $OUT/passB.log | cut -c1-300
python3 tools/pmc_mfma_summary.py /tmp/pmcm_A /tmp/pmcm_B $OUT/summary.json "python3 bench.py --steps 2 --warmup 4 --no_cpu_baseline --no_roofline --no_graph --no_literal $*"

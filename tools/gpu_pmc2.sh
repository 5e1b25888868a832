#!/bin/bash
# Memory-path counters of one conv shape (SHAPE=ci,co,h,k,s N=batch): texture addresser / L1 / L2 busy and stall cycles,
# L1->L2 read latency, LDS FIFO stalls.  Separate --pmc passes (no trace domains besides kernel-trace).
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
cd /tmp
for SET in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCP_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_REQ_sum" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  timeout 60 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmc2_$TAG -- python3 $GRAFT_REPO_ROOT/tools/conv_one.py > $GRAFT_REPO_ROOT/gpurun_out/pmc2/$TAG.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/tmp/pmc2_*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(f"{k:40s} n={len(v)} median={sorted(v)[len(v)//2]:.5g}")
PY

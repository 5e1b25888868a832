#!/bin/bash
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmc_$TAG -- python3 $GRAFT_REPO_ROOT/tools/conv_one.py > $GRAFT_REPO_ROOT/gpurun_out/pmc/$TAG.log 2>&1
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc/$TAG.log | cut -c1-200
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/tmp/pmc_*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(f"{k:32s} n={len(v)} median={sorted(v)[len(v)//2]:.5g}")
PY

#!/bin/bash
# HBM traffic counters for the bench step (separate --pmc passes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
mkdir -p gpurun_out/pmcb
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcb_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 4 --no_cpu_baseline --no_roofline --no_graph > $GRAFT_REPO_ROOT/gpurun_out/pmcb/$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmcb_all; cp -r /tmp/pmcb_FETCH_SIZE /tmp/pmcb_all/; cp -r /tmp/pmcb_WRITE_SIZE /tmp/pmcb_all/
python tools/pmc_summary.py /tmp/pmcb_all gpurun_out/pmcb/summary.json

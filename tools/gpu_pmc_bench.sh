#!/bin/bash
# HBM traffic counters for the bench step (separate --pmc passes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
#   bash tools/gpu_pmc_bench.sh [TAG [bench.py args...]]   -> gpurun_out/pmcb_TAG/summary.json
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT
TAG=${1:-r18}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcb_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcb_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 4 --no_cpu_baseline --no_roofline --no_graph "$@" > $OUT/$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmcb_all; mkdir -p /tmp/pmcb_all; cp -r /tmp/pmcb_FETCH_SIZE /tmp/pmcb_all/; cp -r /tmp/pmcb_WRITE_SIZE /tmp/pmcb_all/
python3 tools/pmc_summary.py /tmp/pmcb_all $OUT/summary.json "python3 bench.py --steps 2 --warmup 4 --no_cpu_baseline --no_roofline --no_graph $*"

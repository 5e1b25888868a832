#!/bin/bash
# rocprofv3 kernel stats of the default bench (NHWC) -> gpurun_out/prof_$1
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT
TAG=${1:-x}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_roofline ${@:2} > $GRAFT_REPO_ROOT/gpurun_out/bench_prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$TAG -name "*kernel_trace*" -delete
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_$TAG.csv \;
grep "^{" gpurun_out/bench_prof_$TAG.log | tail -1 | cut -c1-300

#!/bin/bash
# the whole -m gpu suite + smoke, as the driver runs them
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05_full; mkdir -p $OUT; cd $R
timeout 2400 python3 -m pytest tests/ -x -q -m gpu > $OUT/gpu_suite.txt 2>&1; tail -15 $OUT/gpu_suite.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt

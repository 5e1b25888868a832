#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05d; mkdir -p $OUT; cd $R
timeout 1200 python3 -m pytest tests/test_conv_gpu.py -x -q -k "in_launch or both_batchnorm" 2>&1 | tail -15
timeout 1200 python3 -m pytest tests/test_train_step_gpu.py -x -q -k "in_launch" 2>&1 | tail -15
for GB in 2 1 0 2 1 0; do
AFAN_GRID_BN=$GB timeout 600 python3 bench.py --no_cpu_baseline --no_literal --no_roofline --steps 30 > $OUT/r18_bench_gb$GB.json 2> $OUT/r18_bench.err; python3 -c "
import json;d=json.loads(open('$OUT/r18_bench_gb$GB.json').read().strip().splitlines()[-1]);print('GRID_BN=$GB', d['value'],d['ms_per_step'])"
done

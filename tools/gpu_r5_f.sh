#!/bin/bash
# same-box A/B: previous build (tools/probe/_bin/libafan_hip_prev.so) against the tree's library, per workload
#   bash tools/gpu_r5_f.sh [r18 dl101 frcnn r50]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05f; mkdir -p $OUT; cd $R
WHICH=${@:-r18}
timeout 900 python3 -m pytest tests/test_conv_gpu.py tests/test_blocks_gpu.py -x -q 2>&1 | tail -3
declare -A ARGS=( [r18]="--steps 30" [dl101]="--arch deeplabv3plus_resnet101 --steps 12 --warmup 4" [frcnn]="--arch fasterrcnn_resnet101 --steps 10 --warmup 3"
                  [r50]="--arch resnet50 --batch 64 --pgd_steps 3 --steps 10 --warmup 4" )
for W in $WHICH; do
for rep in 1 2; do
for L in prev new; do
  if [ $L = prev ]; then export AFAN_HIP_LIB=$R/tools/probe/_bin/libafan_hip_prev.so; else unset AFAN_HIP_LIB; fi
  timeout 900 python3 bench.py --no_cpu_baseline --no_literal --no_roofline ${ARGS[$W]} > $OUT/bench_${W}_$L.json 2> $OUT/bench.err; python3 -c "
import json;d=json.loads(open('$OUT/bench_${W}_$L.json').read().strip().splitlines()[-1]);print('$W $L', d['value'],d['ms_per_step'])"
done; done; done
unset AFAN_HIP_LIB

#!/bin/bash
# same-box A/B of an environment switch: AB_ENV="NAME=value" (the "off" setting) against the default, per workload
#   AB_ENV=AFAN_GRID_BN_1X1=0 bash tools/gpu_r5_i.sh [r18 dl101 frcnn r50]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05i; mkdir -p $OUT; cd $R
WHICH=${@:-r50}
declare -A ARGS=( [r18]="--steps 30" [dl101]="--arch deeplabv3plus_resnet101 --steps 12 --warmup 4" [frcnn]="--arch fasterrcnn_resnet101 --steps 10 --warmup 3"
                  [r50]="--arch resnet50 --batch 64 --pgd_steps 3 --steps 10 --warmup 4" )
for W in $WHICH; do
for rep in 1 2; do
for L in off on; do
  if [ $L = off ]; then E="$AB_ENV"; else E="_AB_NONE=1"; fi
  env $E timeout 900 python3 bench.py --no_cpu_baseline --no_literal --no_roofline ${ARGS[$W]} > $OUT/bench_${W}_$L.json 2> $OUT/bench.err; python3 -c "
import json;d=json.loads(open('$OUT/bench_${W}_$L.json').read().strip().splitlines()[-1]);print('$W $L', d['value'],d['ms_per_step'],d['config'].get('final_loss'))"
done; done; done

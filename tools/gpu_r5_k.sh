#!/bin/bash
# one workload under several settings of ONE environment variable, interleaved twice:  VAR=AFAN_CONV_EPI_PF VALS="0 1 3 5 7" bash tools/gpu_r5_k.sh r18
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05kk; mkdir -p $OUT; cd $R
W=${1:-r18}
declare -A ARGS=( [r18]="--steps 30" [dl101]="--arch deeplabv3plus_resnet101 --steps 12 --warmup 4" [frcnn]="--arch fasterrcnn_resnet101 --steps 10 --warmup 3"
                  [r50]="--arch resnet50 --batch 64 --pgd_steps 3 --steps 10 --warmup 4" )
for rep in 1 2; do for V in $VALS; do
  env $VAR=$V timeout 900 python3 bench.py --no_cpu_baseline --no_literal --no_roofline ${ARGS[$W]} > $OUT/b.json 2> $OUT/bench.err; python3 -c "
import json;d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]);print('$W $VAR=$V', d['value'],d['ms_per_step'])"
done; done

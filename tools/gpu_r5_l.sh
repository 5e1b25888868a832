#!/bin/bash
# several builds of the library on one box, interleaved twice:  LIBS="nopf v2 tree" [ENVS="A=1"] bash tools/gpu_r5_l.sh r18   (tools/probe/_bin/libafan_hip_<name>.so; tree = the working tree's)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05l; mkdir -p $OUT; cd $R
W=${1:-r18}
declare -A ARGS=( [r18]="--steps 30" [dl101]="--arch deeplabv3plus_resnet101 --steps 12 --warmup 4" [frcnn]="--arch fasterrcnn_resnet101 --steps 10 --warmup 3"
                  [r50]="--arch resnet50 --batch 64 --pgd_steps 3 --steps 10 --warmup 4" )
for rep in 1 2; do for L in $LIBS; do
  if [ $L = tree ]; then unset AFAN_HIP_LIB; else export AFAN_HIP_LIB=$R/tools/probe/_bin/libafan_hip_$L.so; fi
  env ${ENVS:-_X=1} timeout 900 python3 bench.py --no_cpu_baseline --no_literal --no_roofline ${ARGS[$W]} > $OUT/b.json 2> $OUT/bench.err; python3 -c "
import json;d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]);print('$W $L', d['value'],d['ms_per_step'])"
done; done
unset AFAN_HIP_LIB

#!/bin/bash
# same-box A/B of environment switches over the four benches: tools/ab_env.sh "<env A>" "<env B>" ...  (each argument one setting,
# "" = the defaults); two interleaved rounds; prints ms per step
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
run() { env $1 python3 bench.py $2 --no_dp_schedule --no_literal --no_cpu_baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-44s %-34s %8.3f ms  loss %s' % ('$1', '$3', d['ms_per_step'], d['config'].get('final_loss')))"; }
for round in 1 2; do
  for e in "$@"; do
    run "$e" "--arch fasterrcnn_resnet101 --steps 20" frcnn
    run "$e" "--arch deeplabv3plus_resnet101 --batch 2 --pgd_steps 3 --steps 10 --warmup 4" deeplab
    run "$e" "--arch resnet50 --batch 64 --steps 10 --warmup 4" resnet50
    run "$e" "--steps 30 --warmup 10" resnet18
  done
done

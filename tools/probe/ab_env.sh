#!/bin/bash
# A/B of one environment knob on the two main workloads: usage  ab_env.sh VAR v1 v2 ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
VAR=$1; shift
for v in "$@" "$1"; do
  for W in "" "--arch deeplabv3plus_resnet101 --steps 12"; do
    env $VAR=$v python3 bench.py --no_cpu_baseline --no_roofline --steps 30 --warmup 8 $W 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['metric'][:60], d['ms_per_step'], 'ms')"
  done
done

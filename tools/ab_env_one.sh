#!/bin/bash
# same-box A/B of environment switches on ONE bench: tools/ab_env_one.sh "<bench.py arguments>" <rounds> "<env A>" "<env B>" ...
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
ARGS=$1; N=$2; shift 2
for round in $(seq $N); do
  for e in "$@"; do
    env $e python3 bench.py $ARGS --no_dp_schedule --no_literal --no_cpu_baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-50s %8.3f ms  conv kernels %s ms  loss %s' % ('$e', d['ms_per_step'], (d.get('conv_mfma') or {}).get('conv_kernel_ms_per_step_eager'), d['config'].get('final_loss')))"
  done
done

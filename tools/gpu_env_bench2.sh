#!/bin/bash
# usage: gpu_env_bench2.sh "<bench args>" VAR v1 v2 ...
export TMPDIR=/tmp
ARGS=$1; VAR=$2; shift; shift
for V in "$@"; do
  env $VAR=$V timeout 600 python bench.py $ARGS --steps 20 --warmup 6 --no_cpu_baseline 2>&1 | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$VAR=$V', d['value'], 'img/s', d['ms_per_step'], 'ms/step'); 
for k in ('conv_igemm_fwd_kernel','conv_igemm_dgrad_kernel'): print('   ', k, d['kernels'][k])"
done

#!/bin/bash
# usage: gpu_env_bench.sh VAR v1 v2 ... : default bench once per value of the environment variable
export TMPDIR=/tmp
VAR=$1; shift
for V in "$@"; do
  env $VAR=$V timeout 600 python bench.py --steps 20 --warmup 6 --no_cpu_baseline 2>&1 | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$VAR=$V', d['value'], 'img/s', d['ms_per_step'], 'ms/step'); 
for k in ('conv_igemm_fwd_kernel','conv_igemm_dgrad_kernel'): print('   ', k, d['kernels'][k])"
done

#!/bin/bash
export TMPDIR=/tmp
for D in 1 2; do
  AFAN_CONV_DEEP=$D timeout 600 python bench.py --steps 20 --warmup 6 --no_cpu_baseline 2>&1 | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('deep=$D', d['value'], 'img/s', d['ms_per_step'], 'ms/step'); 
for k in ('conv_igemm_fwd_kernel','conv_igemm_dgrad_kernel'): print('   ', k, d['kernels'][k])"
done

#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_train_step_gpu.py tests/test_conv_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -8
for F in "" "--no_batch_final"; do
  timeout 600 python bench.py --steps 20 --warmup 6 --no_cpu_baseline $F 2>&1 | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('flag=[$F]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', 'loss', d['config']['final_loss'])
for k in ('conv_igemm_fwd_kernel','conv_igemm_dgrad_kernel','conv_wgrad_kernel','bn_nhwc_apply_kernel'): print('   ', k, d['kernels'][k])"
done

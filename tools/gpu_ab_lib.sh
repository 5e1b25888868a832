#!/bin/bash
# A/B of two builds of the library on ONE box: cv_a-fan_amd/libafan_hip_base.so (baseline) vs libafan_hip.so (candidate)
export TMPDIR=/tmp
cp cv_a-fan_amd/libafan_hip.so /tmp/cand.so
for L in base cand base cand; do
  if [ $L = base ]; then cp cv_a-fan_amd/libafan_hip_base.so cv_a-fan_amd/libafan_hip.so; else cp /tmp/cand.so cv_a-fan_amd/libafan_hip.so; fi
  timeout 600 python bench.py --steps 20 --warmup 6 --no_cpu_baseline --no_roofline $EXTRA 2>&1 | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$L', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
done
cp /tmp/cand.so cv_a-fan_amd/libafan_hip.so

#!/bin/bash
# same-box A/B: previous build (tools/probe/_bin/libafan_hip_prev.so) against the tree's library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05f; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -3
AFAN_HIP_LIB=$R/tools/probe/_bin/libafan_hip_stamp.so timeout 300 python3 tools/probe/conv_stamps.py > $OUT/conv_stamps.txt 2>&1; grep "^==\|MFMA wave\|producer:" $OUT/conv_stamps.txt
for rep in 1 2; do
for L in prev new; do
  if [ $L = prev ]; then export AFAN_HIP_LIB=$R/tools/probe/_bin/libafan_hip_prev.so; else unset AFAN_HIP_LIB; fi
  timeout 600 python3 bench.py --no_cpu_baseline --no_literal --no_roofline --steps 30 > $OUT/bench_$L.json 2> $OUT/bench.err; python3 -c "
import json;d=json.loads(open('$OUT/bench_$L.json').read().strip().splitlines()[-1]);print('$L', d['value'],d['ms_per_step'])"
done; done
unset AFAN_HIP_LIB
timeout 900 python3 -m pytest tests/test_conv_gpu.py tests/test_train_step_gpu.py -x -q -k "in_launch or both_batchnorm" 2>&1 | tail -3

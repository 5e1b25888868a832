#!/bin/bash
# deep-pipeline conv variants: correctness under AFAN_CONV_DEEP=4 and 3, then per-layer A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
for D in 4 3; do
AFAN_CONV_DEEP=$D timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -2
done
for D in 0 3 4; do
  echo "AFAN_CONV_DEEP=$D"
  AFAN_CONV_DEEP=$D NO_MIOPEN=1 timeout 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "^ci" | sed 's/miopen[^|]*//g'
done

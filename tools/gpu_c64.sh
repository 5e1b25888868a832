#!/bin/bash
# c64 conv kernel: correctness tests, then A/B microbench (generic kernel vs weights-in-registers kernel)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids > gpurun_out/pytest_conv.log
tail -2 gpurun_out/pytest_conv.log; grep -n "^E  " gpurun_out/pytest_conv.log | head -20
for C in 0 1; do
  for N in 64 256 512; do
  echo "AFAN_CONV_C64=$C N=$N"
  N=$N AFAN_CONV_C64=$C NO_MIOPEN=1 ONLY_FIRST=1 timeout 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "ci  64 co  64" | sed 's/miopen[^|]*//g'
  done
done

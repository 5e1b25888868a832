#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | grep -v "c64 stamps" | tail -3
N=256 AFAN_CONV_C64=1 NO_MIOPEN=1 ONLY_FIRST=1 timeout 300 python tools/conv_bench.py 2>&1 | grep "c64 stamps\|ci  64 co  64" | cut -c1-200 | tail -4

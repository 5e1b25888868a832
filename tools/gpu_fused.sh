#!/bin/bash
export TMPDIR=/tmp
FUSED=1 NO_MIOPEN=1 timeout 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "^ci\|fused" | sed 's/miopen[^|]*//g'

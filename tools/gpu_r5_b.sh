#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05b; mkdir -p $OUT; cd $R
timeout 120 tools/probe/_bin/gb2 > $OUT/gb2.txt 2>&1; cat $OUT/gb2.txt
AFAN_HIP_LIB=$R/tools/probe/_bin/libafan_hip_stamp.so timeout 300 python3 tools/probe/conv_stamps.py > $OUT/conv_stamps.txt 2>&1; cat $OUT/conv_stamps.txt
timeout 300 python3 -m pytest tests/test_det_model_gpu.py -x -q -k "freed" 2>&1 | tail -15

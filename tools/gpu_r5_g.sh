#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05g; mkdir -p $OUT; cd $R
CONV_STAMPS=${CONV_STAMPS:-all} AFAN_HIP_LIB=$R/tools/probe/_bin/libafan_hip_stamp.so timeout 600 python3 tools/probe/conv_stamps.py > $OUT/conv_stamps.txt 2>&1; grep -v "^   *[0-9]*:" $OUT/conv_stamps.txt | tail -40

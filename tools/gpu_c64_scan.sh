#!/bin/bash
export TMPDIR=/tmp
for N in 32 64 128 256 512; do
  echo "N=$N"
  N=$N AFAN_CONV_C64=1 NO_MIOPEN=1 ONLY_FIRST=1 timeout 300 python tools/conv_bench.py 2>&1 | grep "ci  64 co  64" | cut -c1-60
done

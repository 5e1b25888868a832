#!/bin/bash
# Diagnostic builds of the tiled convolution kernel (afan_conv.hip only; the other objects are the product's).
#   tools/build_ablate.sh NAME [-Dflag ...]   ->  cv_a-fan_amd/exp/libafan_hip_NAME.so, selected with AFAN_HIP_LIB=...
# Flags: -DAFAN_CONV_ABLATE=n removes one part of the K loop (results are garbage; only the time means something):
#   1 no operand DMA in the loop, 2 no LDS reads / MFMA, 3 MFMA on constant fragments (no LDS reads), 4 no DMA and no
#   barriers, 5 no K loop (prologue + epilogue), 6 no epilogue, 7 empty kernel;  -DAFAN_CONV_FRAG_BATCH=1|2|4.
# Without arguments: abl1 .. abl7.
set -e
cd "$(dirname "$0")/../cv_a-fan_amd/csrc"
make -s
mkdir -p ../exp
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math"
OTHERS=$(ls *.o | grep -v '^afan_conv\.o$')
one() {
  name=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c afan_conv.hip -o ../exp/afan_conv_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS ../exp/afan_conv_$name.o -o ../exp/libafan_hip_$name.so
  echo "built exp/libafan_hip_$name.so"
}
if [ $# -eq 0 ]; then
  for n in 1 2 3 4 5 6 7; do one abl$n -DAFAN_CONV_ABLATE=$n; done
else
  one "$@"
fi

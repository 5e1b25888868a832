#!/bin/bash
# libafan_hip_prev.so for same-box A/Bs (tools/gpu_r6.sh libs): the library of a COMMITTED tree (default HEAD), built out of tree
#   bash tools/build_prev.sh [rev]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:-HEAD}
D=/tmp/afan_prev_build; rm -rf $D; mkdir -p $D
git -C $R archive $REV "cv_a-fan_amd/csrc" include | tar -x -C $D
make -C "$D/cv_a-fan_amd/csrc" -j6 > /dev/null
mkdir -p $R/tools/probe/_bin
cp "$D/cv_a-fan_amd/libafan_hip.so" $R/tools/probe/_bin/libafan_hip_prev.so
ls -la $R/tools/probe/_bin/libafan_hip_prev.so

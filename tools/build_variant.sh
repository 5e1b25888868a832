#!/bin/bash
# The working tree's library with extra compile flags, built out of tree -> tools/probe/_bin/libafan_hip_NAME.so (for tools/gpu_r6.sh libs)
#   bash tools/build_variant.sh pprio "-DAFAN_HL_PPRIO=3"
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=/tmp/afan_var_$1; rm -rf $D; mkdir -p $D/pkg
cp -r $R/cv_a-fan_amd/csrc $D/pkg/csrc; cp -r $R/include $D/include
rm -f $D/pkg/csrc/*.o
make -C $D/pkg/csrc -j6 EXTRA="$2" > /dev/null
mkdir -p $R/tools/probe/_bin
cp $D/pkg/libafan_hip.so $R/tools/probe/_bin/libafan_hip_$1.so; ls -la $R/tools/probe/_bin/libafan_hip_$1.so

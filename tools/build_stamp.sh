#!/bin/bash
# diagnostic build: libafan_hip_stamp.so = the library with afan_conv.hip compiled -DAFAN_CONV_STAMP (per-tap cycle stamps of the
# halo-form convolution; tools/probe/conv_stamps.py reads them).  Not shipped, not loaded by the product (AFAN_HIP_LIB selects it).
set -e
cd "$(dirname "$0")/../cv_a-fan_amd/csrc"
make -j4 > /dev/null
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DAFAN_CONV_STAMP"
/opt/rocm/bin/hipcc $F -c afan_conv.hip -o /tmp/afan_conv_stamp.o &
rm -f /tmp/afan_conv_bnf_stamp.o
/opt/rocm/bin/hipcc $F -c afan_conv_bnf.hip -o /tmp/afan_conv_bnf_stamp.o 2> /tmp/afan_conv_bnf_stamp.err &
wait
# (the stamped in-launch-BatchNorm unit trips a code generator error in some states of the kernel — "Illegal instruction detected ...
# src_shared_base"; the shipped object then stands in and the probe skips that unit's tables)
[ -f /tmp/afan_conv_bnf_stamp.o ] || { echo "stamped afan_conv_bnf.hip did not compile; using the plain object"; cp afan_conv_bnf.o /tmp/afan_conv_bnf_stamp.o; }
OBJS=$(ls *.o | grep -v '^afan_conv.o$' | grep -v '^afan_conv_bnf.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/afan_conv_stamp.o /tmp/afan_conv_bnf_stamp.o -o ../../tools/probe/_bin/libafan_hip_stamp.so
ls -la ../../tools/probe/_bin/libafan_hip_stamp.so

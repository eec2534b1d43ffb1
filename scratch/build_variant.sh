#!/bin/bash
# build_variant.sh <conv source> <tag>: libvipe_amd with another conv_mfma.hip (same-box A/B of kernel variants via VIPE_AMD_LIB)
set -e
cd /root/repo
SRC=$1; TAG=$2
cp $SRC vipe_amd/csrc/_ab_conv.hip.tmp
mkdir -p scratch/lib
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wno-unused-result -munsafe-fp-atomics"
# compile from inside csrc so that relative includes resolve
(cd vipe_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -x hip -c _ab_conv.hip.tmp -o /root/repo/scratch/lib/conv_$TAG.o)
rm vipe_amd/csrc/_ab_conv.hip.tmp
OBJS=$(ls vipe_amd/lib/obj/*.o | grep -v conv_mfma)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib/libvipe_$TAG.so $OBJS scratch/lib/conv_$TAG.o
echo built scratch/lib/libvipe_$TAG.so

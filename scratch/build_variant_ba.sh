#!/bin/bash
# build_variant_ba.sh <ba source> <tag> [extra hipcc flags]: libvipe_amd with another ba.hip (same-box A/B via VIPE_AMD_LIB)
set -e
cd /root/repo
SRC=$1; TAG=$2; shift 2
cp $SRC vipe_amd/csrc/_ab_ba.hip.tmp
mkdir -p scratch/lib
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wno-unused-result -munsafe-fp-atomics"
(cd vipe_amd/csrc && /opt/rocm/bin/hipcc $FLAGS "$@" -I/root/repo/include -x hip -c _ab_ba.hip.tmp -o /root/repo/scratch/lib/ba_$TAG.o)
rm vipe_amd/csrc/_ab_ba.hip.tmp
OBJS=$(ls vipe_amd/lib/obj/*.o | grep -v "/ba\.")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib/libvipe_$TAG.so $OBJS scratch/lib/ba_$TAG.o
echo built scratch/lib/libvipe_$TAG.so

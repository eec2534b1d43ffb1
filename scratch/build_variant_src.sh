#!/bin/bash
# build_variant_src.sh <source basename in vipe_amd/csrc, e.g. corr_lookup.hip> <tag> [extra hipcc flags...]:
# libvipe_amd with ONE source recompiled under extra flags (same-box A/B via VIPE_AMD_LIB=scratch/lib/libvipe_<tag>.so)
set -e
cd /root/repo
SRC=$1; TAG=$2; shift 2
mkdir -p scratch/lib
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wno-unused-result -munsafe-fp-atomics"
(cd vipe_amd/csrc && /opt/rocm/bin/hipcc $FLAGS "$@" -c $SRC -o /root/repo/scratch/lib/${SRC}_$TAG.o)
OBJS=$(ls vipe_amd/lib/obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib/libvipe_$TAG.so $OBJS scratch/lib/${SRC}_$TAG.o
echo built scratch/lib/libvipe_$TAG.so

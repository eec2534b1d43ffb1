#!/bin/bash
# ablation builds of the conv kernels: scratch/lib/libvipe_amd_abl_<tag>.so with the given -D flags
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p scratch/lib
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wno-unused-result -munsafe-fp-atomics "$@" -c vipe_amd/csrc/conv_mfma.hip -o scratch/lib/conv_abl_$tag.o
objs=$(ls vipe_amd/lib/obj/*.o | grep -v conv_mfma)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib/libvipe_amd_abl_$tag.so scratch/lib/conv_abl_$tag.o $objs
echo built $tag

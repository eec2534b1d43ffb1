#!/bin/bash
# diagnostic build: conv kernels with in-kernel phase stamps -> scratch/lib/libvipe_amd_diag.so
set -e
cd "$(dirname "$0")/.."
python -m vipe_amd.build >/dev/null
mkdir -p scratch/lib
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wno-unused-result -munsafe-fp-atomics -DVIPE_CONV_STAMPS -c vipe_amd/csrc/conv_mfma.hip -o scratch/lib/conv_mfma_diag.o
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wno-unused-result -munsafe-fp-atomics -DVIPE_BA_STAMPS -c vipe_amd/csrc/ba.hip -o scratch/lib/ba_diag.o
objs=$(ls vipe_amd/lib/obj/*.o | grep -v "conv_mfma\|/ba.hip")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib/libvipe_amd_diag.so scratch/lib/conv_mfma_diag.o scratch/lib/ba_diag.o $objs
echo built scratch/lib/libvipe_amd_diag.so

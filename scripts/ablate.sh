#!/bin/bash
# Build timing-only ablation variants of the MFMA kernel (see FINC_ABLATE in finc_mfma.hip) into gpurun_out/ablate/.
set -e
cd "$(dirname "$0")/../fincflow_amd/csrc"
mkdir -p ../../ablate_build
for v in 0 1 2 3; do
  hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++20 -mllvm -amdgpu-mfma-vgpr-form -DFINC_ABLATE=$v -c finc_mfma.hip -o ../../ablate_build/mfma_$v.o &
done
wait
for v in 0 1 2 3; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ablate_build/libfinc_v$v.so finc_abi.o finc_generic.o finc_conv.o finc_gradw.o finc_mix.o ../../ablate_build/mfma_$v.o
done
ls -la ../../ablate_build/*.so

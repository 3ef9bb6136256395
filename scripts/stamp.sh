#!/bin/bash
# Build the stamped diagnostic variant of the inverse kernel into ablate_build/libfinc_stamp.so (never shipped).
set -e
cd "$(dirname "$0")/../fincflow_amd/csrc"
mkdir -p ../../ablate_build
hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++20 -I../../include -mllvm -amdgpu-mfma-vgpr-form -DFINC_STAMP -c finc_mfma.hip -o ../../ablate_build/mfma_stamp.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ablate_build/libfinc_stamp.so finc_abi.o finc_generic.o finc_conv.o ../../ablate_build/mfma_stamp.o

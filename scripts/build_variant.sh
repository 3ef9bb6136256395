#!/bin/bash
# Build an experiment variant of the MFMA inverse into ablate_build/libfinc_<name>.so (travels to the GPU box, not to git):
#   scripts/build_variant.sh <name> [-DFLAG ...]
# Only the c3 / c2 kernels are instantiated (-DFINC_ONLY_C3) so a build takes seconds; the other objects are the product's.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../fincflow_amd/csrc"
mkdir -p ../../ablate_build
hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++20 -mllvm -amdgpu-mfma-vgpr-form -DFINC_EXPERIMENT -DFINC_ONLY_C3 "$@" -c finc_mfma.hip -o ../../ablate_build/mfma_$NAME.o &
hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++20 -mllvm -amdgpu-mfma-vgpr-form -DFINC_EXPERIMENT -DFINC_ONLY_C3 "$@" -c finc_split.hip -o ../../ablate_build/split_$NAME.o
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ablate_build/libfinc_$NAME.so finc_abi.o finc_generic.o finc_f64.o finc_chain.o finc_big.o finc_conv.o finc_wino.o finc_gradw.o finc_mix.o finc_probe.o finc_wino5.o finc_wino4m.o finc_stream.o ../../ablate_build/mfma_$NAME.o ../../ablate_build/split_$NAME.o
echo built ablate_build/libfinc_$NAME.so

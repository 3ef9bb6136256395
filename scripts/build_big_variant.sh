#!/bin/bash
# scripts/build_big_variant.sh <name> [-DFLAG ...]: experiment variant of finc_big.hip only -> ablate_build/libfinc_<name>.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../fincflow_amd/csrc"
mkdir -p ../../ablate_build
hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++20 -mllvm -amdgpu-mfma-vgpr-form -DFINC_EXPERIMENT "$@" -c finc_big.hip -o ../../ablate_build/big_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ablate_build/libfinc_$NAME.so finc_abi.o finc_generic.o finc_f64.o finc_chain.o finc_mfma.o finc_split.o finc_conv.o finc_wino.o finc_gradw.o finc_mix.o ../../ablate_build/big_$NAME.o
echo built ablate_build/libfinc_$NAME.so

#!/bin/bash
# scripts/build_wino_variant.sh <name> [-DFLAG ...]: experiment variant of finc_wino.hip only -> ablate_build/libfinc_<name>.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../fincflow_amd/csrc"
mkdir -p ../../ablate_build
hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++20 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -DFINC_EXPERIMENT "$@" -c finc_wino.hip -o ../../ablate_build/wino_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ablate_build/libfinc_$NAME.so finc_abi.o finc_generic.o finc_f64.o finc_chain.o finc_mfma.o finc_split.o finc_big.o finc_conv.o finc_gradw.o finc_mix.o ../../ablate_build/wino_$NAME.o
echo built ablate_build/libfinc_$NAME.so

#!/bin/bash
# Build an experiment variant of the grad-weight kernels into ablate_build/libfinc_gw_<name>.so (travels to the GPU box, not to git):
#   scripts/build_gradw_variant.sh <name> [-DFLAG ...]
set -e
NAME=$1; shift
cd "$(dirname "$0")/../fincflow_amd/csrc"
mkdir -p ../../ablate_build
hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++20 "$@" -c finc_gradw.hip -o ../../ablate_build/gradw_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ablate_build/libfinc_gw_$NAME.so finc_abi.o finc_generic.o finc_f64.o finc_chain.o finc_mfma.o finc_split.o finc_big.o finc_conv.o finc_wino.o finc_mix.o finc_probe.o finc_wino5.o finc_wino4m.o finc_stream.o ../../ablate_build/gradw_$NAME.o
echo built ablate_build/libfinc_gw_$NAME.so

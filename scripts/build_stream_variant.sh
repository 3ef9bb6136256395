#!/bin/bash
# Build an experiment variant of the streaming-bank kernels into ablate_build/libfinc_<name>.so (travels to the GPU box, not to git):
#   scripts/build_stream_variant.sh <name> [-DFINC_STREAM_ABLATE=1 | -DFINC_STREAM_SPD=12 ...]
# Only finc_stream.hip is recompiled; the other objects are the product's (run `make` first).  Select with FINCFLOW_LIB.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../fincflow_amd/csrc"
mkdir -p ../../ablate_build
hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++20 -mllvm -amdgpu-mfma-vgpr-form -DFINC_EXPERIMENT "$@" -c finc_stream.hip -o ../../ablate_build/stream_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ablate_build/libfinc_$NAME.so finc_abi.o finc_generic.o finc_mfma.o finc_split.o finc_f64.o finc_chain.o finc_big.o finc_conv.o finc_wino.o finc_gradw.o finc_mix.o finc_probe.o finc_wino5.o finc_wino4m.o ../../ablate_build/stream_$NAME.o
echo built ablate_build/libfinc_$NAME.so

#!/bin/bash
# Pricing builds of the library (ICN_EXP bits, csrc/icn_kernels.hip): tools/build_exp.sh <bits> [<bits> ...]
#   -> geniconet_amd/csrc/build_exp/libicn_exp<bits>.so   (git-ignored; travels to the GPU box; load with ICN_LIB_PATH)
# Only icn_kernels.hip is recompiled; the other objects come from the regular build (run `make -C geniconet_amd/csrc` first).
set -e
cd "$(dirname "$0")/../geniconet_amd/csrc"
mkdir -p build_exp
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-const-variable"
for n in "$@"; do
  # (icn_kernels.hip is built without packed fp32 instructions, as in the Makefile)
  /opt/rocm/bin/hipcc $FLAGS -Xclang -target-feature -Xclang -packed-fp32-ops -DICN_EXP=$n ${EXTRA_DEFS} -c -o build_exp/icn_kernels_$n.o icn_kernels.hip 2> >(grep -v "is not a recognized feature for this target" >&2) &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -shared -o build_exp/libicn_exp$n.so build_exp/icn_kernels_$n.o build/icn_api.o build/icn_geometry.o build/icn_bn.o build/icn_loss.o build/icn_optim.o
done
ls -la build_exp/*.so

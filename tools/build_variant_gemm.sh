#!/bin/bash
# tools/build_variant_gemm.sh <name> <-D flags...>: as build_variant.sh, for gemm_conv.hip (the GEMM epilogues)
set -e
name=$1; shift
cd "$(dirname "$0")/../ml-mdm_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c gemm_conv.hip -o /tmp/gemm_conv_$name.o
objs=$(ls build/*.o | grep -v gemm_conv)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../mdm_hip/lib_$name.so /tmp/gemm_conv_$name.o $objs
echo built lib_$name.so

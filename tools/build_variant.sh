#!/bin/bash
# tools/build_variant.sh <name> <-D flags...>: a second libmdm_hip built with other macros for attention.hip (development A/B
# inside one gpurun call: MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_<name>.so python ...); the other objects come from the regular build
set -e
name=$1; shift
cd "$(dirname "$0")/../ml-mdm_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c attention.hip -o /tmp/attention_$name.o
objs=$(ls build/*.o | grep -v attention)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../mdm_hip/lib_$name.so /tmp/attention_$name.o $objs
echo built lib_$name.so

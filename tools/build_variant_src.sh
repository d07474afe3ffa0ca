#!/bin/bash
# tools/build_variant_src.sh <name> <source.hip> <-D flags...>: a second libmdm_hip with ONE source rebuilt under other macros
# (development A/B inside one gpurun call: MDM_HIP_LIB=ml-mdm_amd/mdm_hip/lib_<name>.so python ...)
set -e
name=$1; src=$2; shift; shift
cd "$(dirname "$0")/../ml-mdm_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $src -o /tmp/${src}_$name.o
objs=$(ls build/*.o | grep -v "build/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../mdm_hip/lib_$name.so /tmp/${src}_$name.o $objs
echo built lib_$name.so

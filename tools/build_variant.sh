#!/bin/bash
# tools/libs/libcnr_<name>.so: the library with ONE translation unit rebuilt with extra flags (an A/B arm for
# tools/ab_step.py: CNR_HIP_LIB=tools/libs/libcnr_<name>.so).  usage: build_variant.sh <name> <file.hip> <flags...>
# (the 8-wave kernel lives in fused_bwd_pipe8_kernel.h: configs[1]'s one-launch instantiation is in fused_bwd_pipe8_w0.hip)
set -e
name=$1; file=$2; shift 2
cd "$(dirname "$0")/../category-nerf-reconstruction-official_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c "$file" -o "../../tools/libs/${name}_${file%.hip}.o"
objs=$(ls build/*.o | grep -v "build/${file%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "../../tools/libs/libcnr_${name}.so" $objs "../../tools/libs/${name}_${file%.hip}.o"
echo "built tools/libs/libcnr_${name}.so"

#!/bin/bash
# builds tools/exp/libs/libcnr_tailst.so: the library with tail.hip's per-block-type time stamps compiled in
# (tools/exp/tail_phases.py); run `make -C category-nerf-reconstruction-official_amd/csrc` first
cd "$(dirname "$0")/../../category-nerf-reconstruction-official_amd/csrc" && mkdir -p ../../tools/exp/libs && \
objs=$(ls build/*.o | grep -v "build/tail.o") && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -DCNR_TAIL_STAMPS -c tail.hip -o build/tail_stamps.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libs/libcnr_tailst.so $objs build/tail_stamps.o

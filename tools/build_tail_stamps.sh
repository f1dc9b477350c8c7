#!/bin/bash
# builds tools/libs/libcnr_stamps.so: the library with the per-job time stamps of the step's first and last launch
# compiled in (tools/tail_phases.py); run `make -C category-nerf-reconstruction-official_amd/csrc` first
cd "$(dirname "$0")/../category-nerf-reconstruction-official_amd/csrc" && mkdir -p ../../tools/libs && \
objs=$(ls build/*.o | grep -v "build/tail.o\|build/fused_fwd.o\|_stamps.o") && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -fno-slp-vectorize -DCNR_TAIL_STAMPS -c tail.hip -o ../../tools/libs/tail_stamps.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -fno-slp-vectorize -DCNR_PREP_STAMPS -c fused_fwd.hip -o ../../tools/libs/fused_fwd_stamps.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libs/libcnr_stamps.so $objs ../../tools/libs/tail_stamps.o ../../tools/libs/fused_fwd_stamps.o

#!/bin/bash
# tools/libs/libcnr_hip_timed.so: the whole library with the 8-wave kernel's cycle stamps compiled in
# (CNR_HIP_LIB=tools/libs/libcnr_hip_timed.so python tools/run_train_timed.py)
cd "$(dirname "$0")/../category-nerf-reconstruction-official_amd/csrc" && mkdir -p ../../tools/libs && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -DCNR_PIPE_STAMPS ${CNR_STAMP_ITER:+-DCNR_STAMP_ITER=$CNR_STAMP_ITER} -shared *.hip -o ../../tools/libs/libcnr_hip_timed${CNR_STAMP_ITER}.so

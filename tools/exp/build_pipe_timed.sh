#!/bin/bash
# builds tools/exp/libpipetimed.so: the pipelined backward with cycle stamps compiled in
cd "$(dirname "$0")/../../category-nerf-reconstruction-official_amd/csrc" && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DCNR_PIPE_STAMPS -shared fused_bwd_pipe.hip fused_bwd_pipe8.hip fused_bwd.hip -o ../../tools/exp/libpipetimed.so

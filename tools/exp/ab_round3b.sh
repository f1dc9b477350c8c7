# same-box A/B of the background step: the separate composite / loss launch (CNR_BG_FUSE_RENDER=0) against cnr_bg_backward_render
set -e
mkdir -p gpurun_out/s2
timeout -k 10 300 python -m pytest tests/test_bg_fused_gpu.py -x -q > gpurun_out/s2/bgtest.log 2>&1 || { tail -30 gpurun_out/s2/bgtest.log; exit 1; }
tail -2 gpurun_out/s2/bgtest.log
for i in 1 2 3; do
  CNR_BG_FUSE_RENDER=0 timeout -k 10 120 python tools/exp/time_bg.py fused 2>&1 | tail -2
  timeout -k 10 120 python tools/exp/time_bg.py fused 2>&1 | tail -2
done

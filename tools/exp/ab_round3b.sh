# same-box A/B of the background step: libs/libcnr_bgbase.so (HEAD's bg_fused.hip) against the tree's library
set -e
mkdir -p gpurun_out/s2
timeout -k 10 300 python -m pytest tests/test_bg_fused_gpu.py -x -q > gpurun_out/s2/bgtest.log 2>&1; tail -2 gpurun_out/s2/bgtest.log
for i in 1 2 3; do
  CNR_HIP_LIB=tools/exp/libs/libcnr_bgbase.so timeout -k 10 120 python tools/exp/time_bg.py fused 2>&1 | tail -2
  timeout -k 10 120 python tools/exp/time_bg.py fused 2>&1 | tail -2
done

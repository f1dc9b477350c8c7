# same-box A/B of the background step: cnr_sample_rays as the first launch of every step (CNR_BG_SAMPLE_IN_TAIL=0) against the sampler inside the previous step's last launch
set -e
mkdir -p gpurun_out/s2
timeout -k 10 300 python -m pytest tests/test_bg_fused_gpu.py -x -q > gpurun_out/s2/bgtest.log 2>&1 || { tail -30 gpurun_out/s2/bgtest.log; exit 1; }
tail -2 gpurun_out/s2/bgtest.log
for i in 1 2 3; do
  CNR_BG_SAMPLE_IN_TAIL=0 timeout -k 10 120 python tools/exp/time_bg.py fused 2>&1 | tail -2
  timeout -k 10 120 python tools/exp/time_bg.py fused 2>&1 | tail -2
done

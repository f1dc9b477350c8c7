# same-box A/B of the background step: CNR_HIP_LIB=<base library> against the tree's library (three alternations), after the
# background tests.  usage: tools/exp/ab_round3b.sh <base .so>
set -e
base=${1:-tools/exp/libs/libcnr_dwbase.so}
mkdir -p gpurun_out/s2
timeout -k 10 300 python -m pytest tests/test_bg_fused_gpu.py -x -q > gpurun_out/s2/bgtest.log 2>&1 || { tail -30 gpurun_out/s2/bgtest.log; exit 1; }
tail -2 gpurun_out/s2/bgtest.log
for i in 1 2 3; do
  CNR_HIP_LIB=$base timeout -k 10 120 python tools/exp/time_bg.py fused 2>&1 | tail -2
  timeout -k 10 120 python tools/exp/time_bg.py fused 2>&1 | tail -2
done

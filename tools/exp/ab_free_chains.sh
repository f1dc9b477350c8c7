set -e
mkdir -p gpurun_out/s2
timeout -k 10 300 python -m pytest tests/test_bg_fused_gpu.py -x -q 2>&1 | tail -2
for i in 1 2; do
  CNR_FULLSTEP_FREE=0 timeout -k 10 150 python tools/exp/time_full.py 1 2>&1 | tail -1
  CNR_FULLSTEP_FREE=1 timeout -k 10 150 python tools/exp/time_full.py 1 2>&1 | tail -1
done
CNR_FULLSTEP_FREE=0 timeout -k 10 150 python tools/exp/time_full.py 1 480 10 2>&1 | tail -1
CNR_FULLSTEP_FREE=1 timeout -k 10 150 python tools/exp/time_full.py 1 480 10 2>&1 | tail -1

#!/bin/bash
# One gpurun call: GPU tests, bench (with cpu_baseline), rocprofv3 kernel stats of the bench, PMC passes (HBM bytes,
# MFMA busy) of the fused kernels.  Results under gpurun_out/$1; tools/refresh_profiles.py copies the summaries.
out=gpurun_out/${1:-round}
mkdir -p $out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
timeout -k 10 600 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-260 $out/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof -o b -- python bench.py --no-cpu-baseline > $out/bench_rocprof.json 2> $out/rocprof.err
mode=${2:-train}   # "train": the trainer's own step (one-launch kernel); "": the stand-alone forward / backward calls
for sz in "2048 64" "8192 128"; do n=$(echo $sz | tr " " x); sz="$sz $mode"
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$n -o p -- python tools/prof_one.py $sz > $out/pmc_fetch_$n.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$n -o p -- python tools/prof_one.py $sz > $out/pmc_write_$n.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq_$n -o p -- python tools/prof_one.py $sz > $out/pmc_sq_$n.log 2>&1
done
timeout -k 10 120 python tools/bench_kernels.py > $out/bench_kernels.log 2>&1; tail -7 $out/bench_kernels.log
ls $out

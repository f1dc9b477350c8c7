#!/bin/bash
# Last collection of a round when only the background chain / the whole-iteration graph changed since tools/collect_round.sh ran:
# GPU tests, the bench line, the background step's kernels and counters, the whole iteration.  -> gpurun_out/<tag>/
tag=${1:-final}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 700 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-200 $out/bench.json
tools/exp/prof_bg.sh ${tag}_bg > $out/bg_kernels.txt 2>&1; cat $out/bg_kernels.txt
bash tools/exp/pmc_bg.sh > $out/bg_pmc.txt 2>&1
python tools/exp/time_full.py 1 > $out/full_iteration.txt 2>&1
python tools/exp/time_full.py 0 >> $out/full_iteration.txt 2>&1
python tools/exp/time_full.py 1 480 10 >> $out/full_iteration.txt 2>&1
CNR_FULLSTEP_FREE=0 python tools/exp/time_full.py 1 >> $out/full_iteration.txt 2>&1
grep -v amdgpu.ids $out/full_iteration.txt

#!/bin/bash
# Second half of a round's collection (after tools/collect_round.sh <tag> train), same gpurun call: the background step's kernels
# and counters, the whole iteration, plain against precise geometry, object counts, and the other bench shapes.  -> gpurun_out/<tag>/
tag=${1:-round}; out=gpurun_out/$tag; mkdir -p $out
tools/exp/prof_bg.sh ${tag}_bg > $out/bg_kernels.txt 2>&1
bash tools/exp/pmc_bg.sh > $out/bg_pmc.txt 2>&1
python tools/exp/time_full.py 1 > $out/full_iteration.txt 2>&1
python tools/exp/time_full.py 0 >> $out/full_iteration.txt 2>&1
python tools/exp/time_full.py 1 480 10 >> $out/full_iteration.txt 2>&1
: > $out/plain_vs_precise.txt
for a in "2048 64 4 0" "2048 64 4 1" "8192 128 4 0" "8192 128 4 1" "2048 64 7 1" "2048 64 12 1"; do python tools/exp/quick_step.py $a >> $out/plain_vs_precise.txt 2>&1; done
python bench.py --classes 16 --no-cpu-baseline --no-extra-legs > $out/bench_c16.json 2> $out/bench_c16.err
python bench.py --rays 8192 --samples 128 --no-cpu-baseline --no-extra-legs > $out/bench_8192x128.json 2> $out/bench_8192x128.err
python bench.py --classes 8 --rays 4096 --samples 128 --latent 32 --no-cpu-baseline --no-extra-legs > $out/bench_c8_4096x128_l32.json 2> $out/bench_c8.err
ls $out

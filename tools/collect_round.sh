#!/bin/bash
# One gpurun call that produces everything profiles/<round>_* holds (tools/refresh_profiles.py gpurun_out/<tag> <round> copies the
# summaries): GPU tests, the bench line (with cpu_baseline), rocprofv3 kernel stats of the same bench command, PMC passes of the
# trainer's own step -- HBM bytes (FETCH_SIZE / WRITE_SIZE, one counter per pass), SQ busy / wait counters, instruction mix --, the
# background step's kernels, the whole iteration, and the bench at the other shapes.
#   usage (from the repo root, on the GPU box):  bash tools/collect_round.sh <tag>
out=gpurun_out/${1:-round}
mkdir -p $out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
timeout -k 10 600 python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-260 $out/bench.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $out/bench_steps20.json 2> $out/bench_steps20.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof -o b -- python3 bench.py --no-cpu-baseline --no-extra-legs > $out/bench_rocprof.json 2> $out/rocprof.err
for sz in "2048 64" "8192 128"; do n=$(echo $sz | tr " " x)
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$n -o p -- python3 tools/prof_one.py $sz train > $out/pmc_fetch_$n.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$n -o p -- python3 tools/prof_one.py $sz train > $out/pmc_write_$n.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq_$n -o p -- python3 tools/prof_one.py $sz train > $out/pmc_sq_$n.log 2>&1
  timeout -k 10 300 bash tools/pmc_mix.sh $sz > $out/pmc_mix_$n.txt 2>&1
done
timeout -k 10 200 bash tools/prof_bg.sh ${1:-round}_bg > $out/bg_kernels.txt 2>&1
timeout -k 10 200 bash tools/pmc_bg.sh > $out/bg_pmc.txt 2>&1
timeout -k 10 120 python tools/time_full.py 1 > $out/full_iteration.txt 2>&1
timeout -k 10 120 python tools/time_full.py 1 480 10 >> $out/full_iteration.txt 2>&1
: > $out/steps.txt
for a in "2048 64 4 0" "2048 64 4 1" "8192 128 4 0" "8192 128 4 1" "2048 64 12 1" "2048 64 31 1" "2048 64 100 1"; do timeout -k 10 120 python tools/quick_step.py $a >> $out/steps.txt 2>&1; done
timeout -k 10 200 python bench.py --classes 16 --no-cpu-baseline --no-extra-legs > $out/bench_c16.json 2> $out/bench_c16.err
timeout -k 10 200 python bench.py --rays 8192 --samples 128 --no-cpu-baseline --no-extra-legs > $out/bench_8192x128.json 2> $out/bench_8192x128.err
timeout -k 10 200 python bench.py --classes 8 --rays 4096 --samples 128 --latent 32 --no-cpu-baseline --no-extra-legs > $out/bench_c8_4096x128_l32.json 2> $out/bench_c8.err
timeout -k 10 300 bash tools/pmc_waits.sh > $out/pmc_waits.txt 2>&1
# (built by hand where they exist: tools/build_timed_lib.sh, hipcc tools/micro/*.hip -o tools/libs/<name>)
[ -f tools/libs/libcnr_hip_timed.so ] && CNR_HIP_LIB=tools/libs/libcnr_hip_timed.so timeout -k 10 120 python tools/run_train_timed.py 2048 64 > $out/stamps_2048x64.txt 2>&1
[ -x tools/libs/mfma_overlap ] && timeout -k 10 60 tools/libs/mfma_overlap > $out/micro_mfma_overlap.txt 2>&1
[ -x tools/libs/mfma_overlap2 ] && timeout -k 10 60 tools/libs/mfma_overlap2 >> $out/micro_mfma_overlap.txt 2>&1
ls $out

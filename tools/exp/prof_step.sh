#!/bin/bash
# kernel durations of the category train step under rocprofv3 (GPU box, repo root): prof_step.sh [tag] [R S n_obj precise]
tag=${1:-step}; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag --output-format csv -- python3 $root/tools/exp/quick_step.py "$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    print(f"{name:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:8.2f} us  {r['Percentage']}%")
PY
tail -1 gpurun_out/prof_$tag.log

#!/bin/bash
# kernel durations of the fused background step under rocprofv3 (run on the GPU box from the repo root): prof_bg.sh [tag] [env=val ...]
tag=${1:-bg}; shift
root=$PWD
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_$tag --output-format csv -- python3 $root/tools/time_bg.py fused > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("bg_", "render_loss", "sample_kernel", "maxdepth", "slice_")):
        short = n.split("(")[1].split(")")[-1] if n.startswith("(anonymous") else n
        name = n.replace("(anonymous namespace)::", "").split("(")[0]
        print(f"{name:28s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:7.2f} us")
        if int(r["Calls"]) >= 500: tot += float(r["AverageNs"]) / 1e3
print("sum of per-step kernels %.1f us" % tot)
PY
grep "background step" gpurun_out/prof_$tag.log

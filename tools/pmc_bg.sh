#!/bin/bash
# SQ counters of the fused background kernels (GPU box, repo root)
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $root/gpurun_out/pmc_bg -o p -- python3 $root/tools/time_bg.py fused > $root/gpurun_out/pmc_bg.log 2>&1
cd $root
python3 - <<'PY'
import csv, collections, glob
f = glob.glob("gpurun_out/pmc_bg/*/p_counter_collection.csv") or glob.glob("gpurun_out/pmc_bg/p_counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if k.startswith("bg_") or "render_loss" in k or "sample" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(f"{k:24s} wave_cycles {wc:12.0f} busy {m.get('SQ_BUSY_CYCLES',0):10.0f} wait_any {m.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:.2f} "
          f"active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} valu {m.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} lds_wait {m.get('SQ_WAIT_INST_LDS',0)/wc:.2f} mfma_busy_cyc {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0):10.0f}")
PY

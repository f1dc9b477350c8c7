#!/bin/bash
# Where the waves of the one-launch step kernel wait (PMC, one pass per group; kernel-trace off).  usage: tools/pmc_waits.sh [R S]
R=${1:-2048}; S=${2:-64}; root=$PWD; out=$root/gpurun_out/pmc_waits_${R}x${S}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/a -o p -- python3 $root/tools/prof_one.py $R $S train > $out/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --output-format csv -d $out/b -o p -- python3 $root/tools/prof_one.py $R $S train > $out/b.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_EXP_CNT SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $out/c -o p -- python3 $root/tools/prof_one.py $R $S train > $out/c.log 2>&1
cd $root
python3 - $out $R $S <<'PY'
import collections, csv, glob, sys
out, R, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = collections.defaultdict(list)
for f in glob.glob(out + "/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "field_bwd_pipe8" in r["Kernel_Name"]:
            m[r["Counter_Name"]].append(float(r["Counter_Value"]))
wc = sum(m["SQ_WAVE_CYCLES"]) / max(len(m["SQ_WAVE_CYCLES"]), 1)
print("cnr_field_train at %d x %d: per launch, and as a fraction of SQ_WAVE_CYCLES" % (R, S))
for k in sorted(m):
    v = sum(m[k]) / len(m[k])
    print("  %-28s %14.0f %8.3f" % (k, v, v / wc if wc else 0))
PY
tail -3 $out/a.log $out/b.log $out/c.log | grep -i "error\|invalid\|not" | head

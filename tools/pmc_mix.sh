#!/bin/bash
# Instruction mix and LDS conflicts of the one-launch step kernel (two PMC passes, kernel-trace off): run on the GPU box from the repo
# root; summary to stdout.  usage: tools/pmc_mix.sh [R S]
R=${1:-2048}; S=${2:-64}; root=$PWD; out=$root/gpurun_out/pmc_mix_${R}x${S}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/a -o p -- python3 $root/tools/prof_one.py $R $S train > $out/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/b -o p -- python3 $root/tools/prof_one.py $R $S train > $out/b.log 2>&1
cd $root
python3 - $out $R $S <<'PY'
import collections, csv, glob, sys
out, R, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = collections.defaultdict(list)
for f in glob.glob(out + "/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "field_bwd_pipe8" in r["Kernel_Name"]:
            m[r["Counter_Name"]].append(float(r["Counter_Value"]))
tiles = R * S / 32
print("cnr_field_train at %d x %d: %d tiles of 32 samples; per launch and per tile (wave-instructions)" % (R, S, tiles))
for k in sorted(m):
    v = sum(m[k]) / len(m[k])
    print("  %-26s %14.0f %10.1f" % (k, v, v / tiles))
PY

"""MFMA busy fraction and wave-state split of the fused kernels from a rocprofv3 --pmc pass (SQ counters).

usage: pmc_sq_summary.py <counter_collection.csv> [more.csv ...]
MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 256 CUs x 4 SIMDs), kernel cycles =
GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs; MI355X_MICROARCH.md)."""
import collections, csv, json, sys

out = {}
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "field_" in k or "reduce_records" in k or "tail_kernel" in k or "param_prep" in k:
            key = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
        m["kernel_cycles"] = cyc
        if cyc:
            m["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024)
        wc = m.get("SQ_WAVE_CYCLES", 0)
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if c in m:
                    m[c + "_frac_of_wave_cycles"] = m[c] / wc
        out[path.split("/")[-2] + ":" + k] = m
print(json.dumps(out, indent=1))

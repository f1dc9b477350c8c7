"""Per-kernel HBM bytes from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each).

usage: pmc_summary.py <fetch_csv> <write_csv> <out_json> R S
FETCH_SIZE / WRITE_SIZE are in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM section; calibrated on
reduce_records_kernel, which reads exactly blocks x 58 368 B): FETCH_SIZE counts 64 B per 128-B request -> x2."""
import csv, json, sys, collections


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
R, S = int(sys.argv[4]), int(sys.argv[5])
out = {"workload": f"tools/prof_one.py {R} {S} (1 class, {R} rays x {S} samples), rocprofv3 --pmc, one counter per "
                   "pass, KB per launch; FETCH_SIZE x2 (64 B counted per 128-B request on gfx950)", "kernels": {}}
want = ("field_fwd_kernel", "field_bwd_pipe8_kernel", "field_bwd_pipe_kernel", "field_bwd_kernel", "reduce_records_kernel",
        "tail_kernel", "param_prep_kernel")
call = 0.0
for name in sorted(set(fetch) | set(write)):
    short = next((w for w in want if w in name), None)
    if not short:
        continue
    key = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    f, w = fetch.get(name, 0.0), write.get(name, 0.0)
    hbm = (2 * f + w) * 1024
    out["kernels"][key] = {"FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w, "hbm_bytes": hbm}
    if short not in ("field_fwd_kernel", "tail_kernel", "param_prep_kernel"):
        call += hbm
    targs = [t.strip() for t in key[key.find("<") + 1:key.rfind(">")].split(",")] if "<" in key else []
    if short == "field_bwd_pipe8_kernel" and len(targs) >= 4 and targs[3] != "0":   # <NCH, NDW, WIDE, KR, TWO, PAD>: KR > 0
        out["cnr_field_train_call_hbm_bytes"] = hbm      # the one-launch step body (KR > 0 instantiation)
        # 12 B pts + 4 B z per sample in; per ray 18 B targets in + 24 B renders out; one 62 KB operand image and one
        # 58 KB record per workgroup are what the launch structure adds
        out["cnr_field_train_algorithmic_bytes"] = R * S * 16 + R * 42
out["cnr_field_bwd_pipe_call_hbm_bytes"] = call
out["cnr_field_bwd_algorithmic_bytes"] = R * S * 28
out["note"] = ("backward call = field kernel + reduce_records; the per-workgroup records (one 59 KB record per workgroup, "
               "written once, read once) dominate; the 28 B/sample of algorithmic input are the rest")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))

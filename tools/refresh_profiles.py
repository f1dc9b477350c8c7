"""Copy the judged summaries of a tools/collect_round.sh run (gpurun_out/<tag>) into profiles/<round>_* :
    python tools/refresh_profiles.py gpurun_out/<tag> r04"""
import csv, json, os, subprocess, sys
src = sys.argv[1]
RND = sys.argv[2] if len(sys.argv) > 2 else "r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
open(os.path.join(P, RND + "_bench_2048x64.json"), "w").write(open(f"{src}/bench.json").read().strip().splitlines()[-1] + "\n")
open(os.path.join(P, RND + "_bench_2048x64_under_rocprof.json"), "w").write(
    open(f"{src}/bench_rocprof.json").read().strip().splitlines()[-1] + "\n")
subprocess.check_call(["cp", f"{src}/rocprof/b_kernel_stats.csv", os.path.join(P, RND + "_bench_2048x64_kernel_stats.csv")])
for n in ("2048x64", "8192x128"):
    for name, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        rows = [r for r in csv.DictReader(open(f"{src}/{d}_{n}/p_counter_collection.csv"))
                if any(k in r["Kernel_Name"] for k in ("field_", "reduce_records", "pack_kernel", "tail_kernel", "param_prep"))]
        with open(os.path.join(P, f"{RND}_pmc_{name}_{n}.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    R, S = n.split("x")
    out = os.path.join(P, RND + "_pmc_traffic.json" if n == "2048x64" else f"{RND}_pmc_traffic_{n}.json")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools/pmc_summary.py"), f"{src}/pmc_fetch_{n}/p_counter_collection.csv",
                           f"{src}/pmc_write_{n}/p_counter_collection.csv", out, R, S], stdout=subprocess.DEVNULL)
# the bench line was printed on the GPU box before this collection's PMC passes existed there: its roofline.traffic is the
# previous collection's figure -> replace it with this collection's (same command, same shape)
bj = os.path.join(P, RND + "_bench_2048x64.json")
line = json.loads(open(bj).read())
call = json.load(open(os.path.join(P, RND + "_pmc_traffic.json")))
key = line["roofline"]["kernel"] + "_call_hbm_bytes"
if key in call:
    line["roofline"]["traffic"] = call[key]
    open(bj, "w").write(json.dumps(line) + "\n")
sq = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools/pmc_sq_summary.py"),
                              f"{src}/pmc_sq_2048x64/p_counter_collection.csv", f"{src}/pmc_sq_8192x128/p_counter_collection.csv"])
open(os.path.join(P, RND + "_pmc_sq_mfma_busy.json"), "wb").write(sq)
for name, dst in (("pmc_mix_2048x64.txt", "_pmc_instruction_mix.txt"), ("pmc_mix_8192x128.txt", "_pmc_instruction_mix_8192x128.txt"),
                  ("bg_kernels.txt", "_bg_step_kernels.txt"), ("bg_pmc.txt", "_bg_step_pmc_sq.txt"),
                  ("full_iteration.txt", "_full_iteration.txt"), ("steps.txt", "_step_times_by_shape.txt"),
                  ("bench_c16.json", "_bench_c16_2048x64.json"), ("bench_8192x128.json", "_bench_8192x128.json"),
                  ("bench_c8_4096x128_l32.json", "_bench_c8_4096x128_l32.json"), ("pytest_gpu.log", "_pytest_gpu_tail.txt"),
                  ("bench_steps20.json", "_bench_2048x64_steps20_warmup5.json"), ("pmc_waits.txt", "_pmc_wait_counters.txt"),
                  ("stamps_2048x64.txt", "_cycle_stamps_2048x64.txt"), ("micro_mfma_overlap.txt", "_micro_mfma_valu_overlap.txt")):
    f = f"{src}/{name}"
    if os.path.exists(f):
        txt = open(f).read()
        if name == "pytest_gpu.log":
            txt = "\n".join(txt.strip().splitlines()[-3:]) + "\n"
        open(os.path.join(P, RND + dst), "w").write(txt)
print("profiles refreshed from", src)

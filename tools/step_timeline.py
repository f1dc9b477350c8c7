"""Print one train step's kernel timeline from a rocprofv3 kernel_trace.csv (start offset, duration, name)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last complete step: from the last maxdepth_kernel but one to the last one
idx = [i for i, n in enumerate(names) if "maxdepth_kernel" in n]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:8.1f} us  +{(e-s)/1e3:6.1f}  q{r.get('Queue_Id','?'):>3}  {r['Kernel_Name'][:70]}")
print("step period", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us")

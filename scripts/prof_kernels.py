"""Per-kernel average duration (us) of a steady-state forward from a rocprofv3 kernel trace: prof_kernels.py <dir> [pattern]"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cuts = [i for i, r in enumerate(rows) if "k_stem" in r["Kernel_Name"] and "fwd" in r["Kernel_Name"]]
body = rows[cuts[1]:cuts[-1]]
steps = len(cuts) - 2
by = {}
for r in body:
    k = r["Kernel_Name"][:60]
    if pat and pat not in k:
        continue
    by.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v) / steps:9.1f} us/step {len(v) / steps:5.1f} calls  {' '.join(f'{x:.0f}' for x in v[:len(v) // steps])}  {k}")
    tot += sum(v) / steps
print(f"{tot:9.1f} us/step total")

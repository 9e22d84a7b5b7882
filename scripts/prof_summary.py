"""Summarise rocprofv3 CSV output.  Usage:
  prof_summary.py stats <dir> <steps> <out.md> <title>        (--kernel-trace --stats: *_kernel_stats.csv)
  prof_summary.py pmc <dir> <passes> <counter>               (--pmc X --kernel-trace: *_counter_collection.csv) -> prints sum/pass
"""
import csv, glob, os, sys


def find(d, suffix):
    f = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not f:
        sys.exit(f"no *{suffix} under {d}")
    return f[-1]


def stats(d, steps, out, title):
    rows = list(csv.DictReader(open(find(d, "kernel_stats.csv"))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write(f"# {title}\n\ntotal kernel time per step: {tot / steps / 1e6:.2f} ms\n\n| kernel | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            t = float(r["TotalDurationNs"])
            if t / tot < 0.002:
                continue
            f.write(f"| `{r['Name'][:70]}` | {int(r['Calls']) / steps:g} | {t / steps / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {100 * t / tot:.1f} |\n")
    print(open(out).read()[:3000])


def pmc(d, passes, counter):
    rows = list(csv.DictReader(open(find(d, "counter_collection.csv"))))
    by = {}
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        by[r["Kernel_Name"][:60]] = by.get(r["Kernel_Name"][:60], 0.0) + float(r["Counter_Value"])
    tot = sum(by.values())
    print(f"{counter}: total {tot:.6g} over {passes} passes -> {tot / passes:.6g} per pass")
    for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {v / passes:14.6g}  {k}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5])
    else:
        pmc(sys.argv[2], int(sys.argv[3]), sys.argv[4])

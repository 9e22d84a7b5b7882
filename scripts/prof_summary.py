"""Summarise rocprofv3 CSV output.  Usage:
  prof_summary.py stats <dir> <steps> <out.md> <title>        (--kernel-trace --stats: *_kernel_stats.csv)
  prof_summary.py steady <dir> <out.md> <title> [steps]       (--kernel-trace: *_kernel_trace.csv; steady-state per-step table)
  prof_summary.py levels <dir> <levels.json> <out.md> <title>  (forward trace + prof_step.py --dump-levels: per resolution level)
  prof_summary.py pmc <dir> <passes> <counter>               (--pmc X --kernel-trace: *_counter_collection.csv) -> prints sum/pass
"""
import csv, glob, os, sys


def find(d, suffix):
    f = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not f:
        sys.exit(f"no *{suffix} under {d}")
    return f[-1]


def steady(d, out, title, nsteps=0):
    """Per-step kernel table from the kernel TRACE, steady state only: the trace is cut at every launch of `marker` (one per
    forward); the first step (module upload = hundreds of __amd_rocclr_copyBuffer launches, kernel attribute setup, plan
    build) is dropped, the rest averaged.  Replaces `stats` for anything quoted per step."""
    rows = list(csv.DictReader(open(find(d, "kernel_trace.csv"))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cuts = [i for i, r in enumerate(rows) if "k_stem" in r["Kernel_Name"] and "fwd" in r["Kernel_Name"]]
    if nsteps and len(cuts) % nsteps == 0 and len(cuts) > nsteps:      # several stem convolutions per forward (variant A)
        cuts = cuts[::len(cuts) // nsteps]
    if len(cuts) < 3:
        sys.exit(f"need >= 3 steps, found {len(cuts)} stem launches")
    # a step begins a few setup launches (memset / dropout masks) before its stem: attribute those to the step they precede
    body = rows[cuts[1]:cuts[-1]]
    steps = len(cuts) - 2
    by = {}
    for r in body:
        k = r["Kernel_Name"][:70]
        t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = by.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += t
    tot = sum(v[1] for v in by.values())
    span = (int(body[-1]["End_Timestamp"]) - int(body[0]["Start_Timestamp"])) / steps
    with open(out, "w") as f:
        f.write(f"# {title}\n\nsteady state, {steps} steps (first step dropped): kernel time {tot / steps / 1e6:.3f} ms per step, "
                f"{sum(v[0] for v in by.values()) / steps:.1f} launches per step, wall span {span / 1e6:.3f} ms per step\n\n"
                "| kernel | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|\n")
        for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
            if t / tot < 0.002:
                continue
            f.write(f"| `{k}` | {n / steps:g} | {t / steps / 1e6:.3f} | {t / n / 1e3:.1f} | {100 * t / tot:.1f} |\n")
    print(open(out).read()[:3500])


def levels(d, levels_json, out, title):
    """Forward time and launch count PER RESOLUTION LEVEL (the map size an op works on: 128, 64, 32, 16, 8): the kernel trace of
    `prof_step.py --fwd-only --dump-levels L.json` is walked in launch order against the plan's op list (every library launch
    must belong to the kernel families of the current or the next op, anything else aborts); torch's own kernels (dropout
    masks) and memsets are listed under `other`.  Steady state: first forward dropped."""
    import json
    ops = json.load(open(levels_json))
    rows = list(csv.DictReader(open(find(d, "kernel_trace.csv"))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cuts = [i for i, r in enumerate(rows) if "k_stem" in r["Kernel_Name"] and "fwd" in r["Kernel_Name"]]
    nstem = sum(1 for o in ops if "k_stem" in o["families"])
    cuts = cuts[::max(1, nstem)]
    if len(cuts) < 3:
        sys.exit("need >= 3 forwards")
    agg, steps = {}, 0
    for a, b in zip(cuts[1:-1], cuts[2:]):
        steps += 1
        j = 0
        for r in rows[a:b]:
            name = r["Kernel_Name"]
            t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            lib = name.startswith("k_") or name.startswith("void k_")
            key = "other"
            if lib:
                base = name.replace("void ", "")
                if j == len(ops) or not any(base.startswith(f) for f in ops[j]["families"]):
                    sys.exit(f"launch {j} of a forward is {name[:50]}, the plan expects {ops[j]['families'] if j < len(ops) else 'nothing'}")
                key = ops[j]["size"]
                j += 1
            e = agg.setdefault(key, [0, 0])
            e[0] += 1
            e[1] += t
    tot = sum(v[1] for v in agg.values())
    with open(out, "w") as f:
        f.write(f"# {title}\n\nper resolution level, steady state ({steps} forwards): {tot / steps / 1e6:.3f} ms kernel time, "
                f"{sum(v[0] for v in agg.values()) / steps:.0f} launches per forward\n\n| map size | launches | ms | avg us | % |\n|---|---|---|---|---|\n")
        for k in sorted(agg, key=lambda k: -(k if isinstance(k, int) else 0)):
            n, t = agg[k]
            f.write(f"| {k} | {n / steps:g} | {t / steps / 1e6:.3f} | {t / n / 1e3:.1f} | {100 * t / tot:.1f} |\n")
    print(open(out).read())


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
    if sys.argv[1] == "steady":
        steady(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 0)
    elif sys.argv[1] == "levels":
        levels(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5])
    else:
        pmc(sys.argv[2], int(sys.argv[3]), sys.argv[4])

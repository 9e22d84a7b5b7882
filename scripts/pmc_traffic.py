"""HBM bytes of one forward plan run from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).
  pmc_traffic.py <fetch_dir> <write_dir> <key e.g. B_bs64_256> <round tag> <out.md>
Both directories hold the `--pmc X --kernel-trace --output-format csv` output of `scripts/prof_step.py --fwd-only --steps K`.
Steady state only: the dispatch list is cut at every stem launch (one per forward); the first forward (plan build, parameter
upload, input synthesis) is dropped and the rest averaged.  Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM
section): both counters are in KB; FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950 and is doubled.
Writes / updates profiles/pmc_traffic.json (keyed entry carrying bench.source_sha16() of the sources that were profiled) and
a per-kernel table."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_forward(d, counter):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    cuts = [i for i, r in enumerate(rows) if "k_stem" in r["Kernel_Name"] and "fwd" in r["Kernel_Name"]]
    nsteps = int(os.environ.get("PMC_STEPS", "4"))
    if len(cuts) % nsteps == 0 and len(cuts) > nsteps:             # several stem convolutions per forward (variant A)
        cuts = cuts[::len(cuts) // nsteps]
    if len(cuts) < 3:
        sys.exit(f"{d}: need >= 3 forwards, found {len(cuts)}")
    body, steps = rows[cuts[1]:cuts[-1]], len(cuts) - 2
    by = {}
    for r in body:
        k = r["Kernel_Name"][:64]
        by[k] = by.get(k, 0.0) + float(r["Counter_Value"]) / steps
    return by, steps


def main():
    fd, wd, key, tag, out = sys.argv[1:6]
    import bench
    fe, s1 = per_forward(fd, "FETCH_SIZE")
    wr, s2 = per_forward(wd, "WRITE_SIZE")
    fkb, wkb = sum(fe.values()), sum(wr.values())
    traffic = (2.0 * fkb + wkb) * 1024.0
    variant = key.split("_")[0]
    alg = bench.ALG_FWD_BYTES[variant] * int(key.split("_bs")[1].split("_")[0])
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        js = json.load(open(path))
    except (OSError, ValueError):
        js = {}
    js["note"] = ("HBM bytes of ONE forward plan run (batch 64, 256x256, train-mode BN): rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                  "in separate passes over scripts/prof_step.py --fwd-only, steady-state forwards only (scripts/pmc_traffic.py); both "
                  "counters are in KB; gfx950 correction: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section).  An entry is quoted by "
                  "bench.py only while src_sha16 equals bench.source_sha16() of the tree being benchmarked.")
    js[key] = {"fetch_kb_per_fwd": round(fkb, 1), "write_kb_per_fwd": round(wkb, 1), "traffic_bytes_per_fwd": round(traffic, 0),
               "algorithmic_bytes_per_fwd": alg, "traffic_over_algorithmic": round(traffic / alg, 3), "forwards_averaged": [s1, s2],
               "round": tag, "src_sha16": bench.source_sha16()}
    json.dump(js, open(path, "w"), indent=1)
    with open(out, "w") as f:
        f.write(f"# HBM traffic of one forward plan run, {key} ({tag})\n\n"
                f"`rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 scripts/prof_step.py --variant {variant} "
                f"--fwd-only --steps K` and the same with WRITE_SIZE; steady-state forwards only ({s1} / {s2} averaged).\n\n"
                f"FETCH_SIZE {fkb:.6g} KB (x2 on gfx950) + WRITE_SIZE {wkb:.6g} KB = **{traffic / 1e9:.2f} GB per forward**; "
                f"algorithmic {alg / 1e9:.2f} GB; ratio {traffic / alg:.2f}; sources {bench.source_sha16()}.\n\n"
                "| kernel | FETCH_SIZE KB (raw) | WRITE_SIZE KB | bytes (2F + W) MB |\n|---|---|---|---|\n")
        for k in sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, 0) + wr.get(k, 0))):
            b = (2 * fe.get(k, 0) + wr.get(k, 0)) * 1024 / 1e6
            if b < 1:
                continue
            f.write(f"| `{k}` | {fe.get(k, 0):.0f} | {wr.get(k, 0):.0f} | {b:.1f} |\n")
    print(open(out).read()[:2500])


if __name__ == "__main__":
    main()

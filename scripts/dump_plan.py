"""Print the forward records of a model's plan (CPU only; no library call).  python scripts/dump_plan.py [A|B] [bs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
from litehandnet_amd import get_model
from litehandnet_amd.config import litehandnet_cfg
from litehandnet_amd.plan import PlanBuilder, STEM, PW, DW, KXK, EW, MAXPOOL, AVGPOOL, CA_MLP, TABLE_FILL, ATT_MLP
variant = sys.argv[1] if len(sys.argv) > 1 else "B"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = get_model(litehandnet_cfg(variant))
tensors = list(m.state_dict(keep_vars=True).values())
pb = PlanBuilder(N, {id(t): j for j, t in enumerate(tensors)}, image_hw=(256, 256), with_backward=False, p_drop=0.0)
y = m.emit(pb, pb.image())
names = {STEM: "stem", PW: "pw", DW: "dw", KXK: "kxk", EW: "ew", MAXPOOL: "maxpool", AVGPOOL: "avgpool", CA_MLP: "ca", TABLE_FILL: "table", ATT_MLP: "ca"}
tot = collections.Counter()
for r in pb.recs:
    k = r["op"]
    if k in (STEM, PW, DW, KXK):
        x, o = r["x"], r["out"]
        nsrc = pb._xs(r)[1] if k in (PW, DW) else 1
        mb = N * 4 * (nsrc * x.H * x.W * x.C + o.H * o.W * o.C) / 1e6
        print(f"{names[k]:8s} in b{x.buf}[{x.coff}:{x.coff + x.C}] {x.H}x{x.W} -> b{o.buf}[{o.coff}:{o.coff + o.C}] {o.H}x{o.W} k{r['k']} s{r['stride']} d{r['dil']} bn={r['bn'] is not None} slope={r['slope']} {mb:.1f}MB")
    elif k == EW and r.get("lazy"):
        print(f"{'ew(lazy)':8s} " + " + ".join(f"{c:g}*b{t.buf}[{t.coff}:{t.coff + t.C}]" for t, c in r["flat"]) + f" -> b{r['out'].buf} (summed on load)")
        continue
    elif k == EW:
        o = r["out"]
        mb = N * 4 * o.H * o.W * o.C * (1 + len(r["srcs"])) / 1e6
        print(f"{'ew':8s} " + " + ".join(f"b{s.buf}[{s.coff}:{s.coff + s.C}]{s.H}x{s.W}" for s in r["srcs"]) + f" -> b{o.buf}[{o.coff}:{o.coff + o.C}] {o.H}x{o.W} slope={r['slope']} {mb:.1f}MB")
    elif k in (CA_MLP, ATT_MLP):
        yv = r["y"]
        mb = N * 4 * yv.H * yv.W * yv.C / 1e6
        print(f"{'ca':8s} gate on b{yv.buf} {yv.H}x{yv.W}x{yv.C} (pool reads {mb:.1f}MB)")
    else:
        x, o = r["x"], r["out"]
        mb = N * 4 * (x.H * x.W * x.C + o.H * o.W * o.C) / 1e6
        print(f"{names[k]:8s} b{x.buf}[{x.coff}:{x.coff + x.C}] {x.H}x{x.W} -> b{o.buf} {o.H}x{o.W} {mb:.1f}MB")
    tot[names[k]] += mb
print({k: round(v) for k, v in tot.items()}, "MB; total", round(sum(tot.values())))

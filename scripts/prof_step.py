"""Profiling driver: a few training steps + forward-only passes of variant B/A at bs64 (run under rocprofv3)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litehandnet_amd import get_loss, get_model, heatmap
from litehandnet_amd.config import litehandnet_cfg
from litehandnet_amd.train import Trainer
ap = argparse.ArgumentParser(); ap.add_argument("--variant", default="B"); ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--fwd-only", action="store_true")
ap.add_argument("--dump-levels", default="", help="write the forward plan's launch list (kernel family, map size per launch) as JSON")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = litehandnet_cfg(a.variant)
m = get_model(cfg).to(dev).train(); crit = get_loss(cfg)
img = torch.randn(a.batch, 3, 256, 256, device=dev)
j = torch.zeros(a.batch, 21, 3, device=dev); j[..., :2] = torch.rand(a.batch, 21, 2, device=dev) * 256
t, w = heatmap.generate_target_batch(j, torch.ones_like(j), [256, 256], [64, 64], 2, True)
if a.fwd_only:
    with torch.no_grad():
        for _ in range(a.steps): m(img)
else:
    tr = Trainer(m, crit)
    for _ in range(a.steps): tr.step(img, {"target": t, "target_weight": w})
torch.cuda.synchronize()
if a.dump_levels:
    import json
    from litehandnet_amd import plan as P
    plan = [p for k, p in m._engine.plans.items() if k[1] == (not a.fwd_only)][0]
    fam = {P.STEM: [["k_stem"]], P.PW: [["k_pw_fwd"]], P.DW: [["k_dwk_fwd", "k_dws2_fwd", "k_dw_fwd"]], P.KXK: [["k_w_tapmajor"], ["k_kxk"]],
           P.EW: [["k_ew_fwd"]], P.MAXPOOL: [["k_maxpool2_fwd"]], P.AVGPOOL: [["k_avgpool", "k_pool_cat"]], P.CA_MLP: [["k_ca1"], ["k_ca2"]],
           P.ATT_MLP: [["k_att1"], ["k_att2"]], P.SHUFFLE: [["k_shuffle2_fwd"]]}
    ops = []          # one entry per expected library launch, in order: allowed kernel-name prefixes + the map size it works on
    cf = plan._keep[1]
    for i in range(plan.n_fwd):
        o = cf[i]
        if o.kind not in fam:
            continue
        b = o.in_buf[0] if (o.kind == P.AVGPOOL or o.out_buf < 0) else o.out_buf
        size = int(plan.pb.bufs[b].H)
        seq = list(fam[o.kind])
        if o.kind in (P.STEM, P.PW, P.DW, P.KXK) and o.p[2] >= 0:
            seq = seq + [["k_bn_finalize"]]
        for fams in seq:
            ops.append({"families": fams, "size": size})
    json.dump(ops, open(a.dump_levels, "w"))
print("done")

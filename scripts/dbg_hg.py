import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import synth, torch_ref
from litehandnet_amd import litehourglass as lh

def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

def run(ours, ref, x, pick=None, seed=0):
    sd = synth.synth_state_dict(ref, seed); ref.load_state_dict(sd); ours.load_state_dict(sd); ours.cuda()
    ref.train(); ours.train()
    xr = x.clone().requires_grad_(); yr = ref(xr)
    if pick is not None: yr = yr[pick]
    g = torch.randn(yr.shape, generator=torch.Generator().manual_seed(1))
    yr.backward(g)
    xg = x.clone().cuda().requires_grad_(); yg = ours(xg); yg.backward(g.cuda())
    rp = dict(ref.named_parameters())
    worst = max((float((p.grad.cpu().double() - rp[k].grad.double()).norm() / (rp[k].grad.double().norm() + 1e-3 * max(float(v.grad.norm()) for v in rp.values()))), k) for k, p in ours.named_parameters())
    print(f"fwd {rel(yg, yr):.2e} dx {rel(xg.grad, xr.grad):.2e} worst param grad {worst[0]:.2e} {worst[1]}")

for (n, c, s, st) in [(2, 64, 32, 4), (4, 64, 64, 4), (8, 64, 64, 4), (4, 64, 64, 2), (4, 64, 32, 1)]:
    print("hourglass N", n, "C", c, "size", s, "stages", st, end=": ")
    run(lh.EncoderDecoder(st, c, "ca", "none", p_drop=0.0), torch_ref._HourglassB(st, c, "ca", "none", 0.0),
        torch.randn(n, c, s, s, generator=torch.Generator().manual_seed(0)), pick=-1)
print("--- rbu ca at small maps")
for (n, c, s) in [(2, 128, 2), (2, 128, 4), (4, 128, 8)]:
    print("rbu N", n, "size", s, end=": ")
    run(lh.RepBasicUnit(c, c, "ca", p_drop=0.0), torch_ref.RepBasicUnit(c, c, "ca", 0.0), torch.randn(n, c, s, s, generator=torch.Generator().manual_seed(0)))
print("--- hourglass with rbu_ca=ca")
for (n, c, s, st) in [(2, 64, 16, 1), (2, 64, 16, 2), (2, 128, 16, 4)]:
    print("hourglass ca/ca N", n, "C", c, "size", s, "stages", st, end=": ")
    run(lh.EncoderDecoder(st, c, "ca", "ca", p_drop=0.0), torch_ref._HourglassB(st, c, "ca", "ca", 0.0),
        torch.randn(n, c, s, s, generator=torch.Generator().manual_seed(0)), pick=-1)

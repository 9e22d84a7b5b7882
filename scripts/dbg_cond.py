import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth, torch_ref
from litehandnet_amd import litehourglass as lh

def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

def go(ours, ref, x, pick=None):
    sd = synth.synth_state_dict(ref, 0); ref.load_state_dict(sd); ours.load_state_dict(sd); ours.cuda(); ours.train(); ref.train()
    ref64 = copy.deepcopy(ref).double()
    g = None
    outs = {}
    for name, m, xx in (("ref64", ref64, x.double()), ("ref32", ref, x)):
        xr = xx.clone().requires_grad_(); y = m(xr)
        if pick is not None: y = y[pick]
        if g is None: g = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
        y.backward(g.to(y.dtype)); outs[name] = (y.detach(), xr.grad)
    xg = x.clone().cuda().requires_grad_(); yg = ours(xg); yg.backward(g.cuda()); outs["ours"] = (yg.detach(), xg.grad)
    print("  fwd: ref32 %.2e ours %.2e | dx: ref32 %.2e ours %.2e" % (rel(outs["ref32"][0], outs["ref64"][0]), rel(outs["ours"][0], outs["ref64"][0]),
          rel(outs["ref32"][1], outs["ref64"][1]), rel(outs["ours"][1], outs["ref64"][1])))

for n in (2, 8, 32):
    for s in (32, 64):
        print("MSRB ca N", n, "size", s)
        go(lh.MSRB(64, 64, "ca", p_drop=0.0), torch_ref.MSRB(64, 64, "ca", 0.0), torch.randn(n, 64, s, s, generator=torch.Generator().manual_seed(0)))
print("hourglass 4 stages N 8 size 64 (ca/none)")
go(lh.EncoderDecoder(4, 64, "ca", "none", p_drop=0.0), torch_ref._HourglassB(4, 64, "ca", "none", 0.0), torch.randn(8, 64, 64, 64, generator=torch.Generator().manual_seed(0)), pick=-1)

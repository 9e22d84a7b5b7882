import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from litehandnet_amd import get_model
from litehandnet_amd.config import litehandnet_cfg
from litehandnet_amd.engine import Engine
from litehandnet_amd import repblocks, litehourglass as lh
from oracle import synth
dev = "cuda"
def rel(a, b): return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
def check(make, x, tag):
    a, b = make(), make()
    sd = synth.synth_state_dict(a, 3)
    a.load_state_dict(sd); b.load_state_dict(sd)
    a.to(dev).train(); b.to(dev).train()
    eng = Engine(b); b.__dict__["_engine"] = eng
    eng.sync_override = (2, lambda t: t.mul_(2))
    xa = x.clone().to(dev).requires_grad_(); xb = x.clone().to(dev).requires_grad_()
    ya = a(xa); yb = b(xb)
    g = torch.randn_like(ya)
    ya.backward(g); yb.backward(g)
    worst = max(rel(pb.grad, pa.grad) for pa, pb in zip(a.parameters(), b.parameters()))
    print(tag, "fwd", rel(yb, ya), "dx", rel(xb.grad, xa.grad) if xa.grad is not None else None, "grads", worst)
r = np.random.Generator(np.random.PCG64(0))
X = lambda n, c, h: torch.from_numpy(r.standard_normal((n, c, h, h)).astype(np.float32))
check(lambda: repblocks.RepConv(64, 64, 1), X(4, 64, 16), "pw")
check(lambda: repblocks.RepConv(64, 64, 3, 1, 1, groups=64, activation=None), X(4, 64, 16), "dw")
check(lambda: lh.RepBasicUnit(64, 64, "none", p_drop=0.0), X(4, 64, 16), "rbu")
check(lambda: lh.RepBasicUnit(64, 64, "ca", p_drop=0.0), X(4, 64, 16), "rbu+ca")
check(lambda: lh.MSRB(64, 64, "ca", p_drop=0.0), X(4, 64, 16), "msrb")
from litehandnet_amd.common import ChannelAttension
check(lambda: ChannelAttension(64, p_drop=0.0), X(4, 64, 16), "ca alone")
# forward-only staged vs whole
a = ChannelAttension(64, p_drop=0.0); b = ChannelAttension(64, p_drop=0.0)
sd = synth.synth_state_dict(a, 3); a.load_state_dict(sd); b.load_state_dict(sd)
a.to(dev).train(); b.to(dev).train()
eng = Engine(b); b.__dict__["_engine"] = eng
calls = []
eng.sync_override = (2, lambda t: (calls.append((t.numel(), t.clone())), t.mul_(2)))
x = X(4, 64, 16).to(dev)
with torch.no_grad():
    ya, yb = a(x), b(x)
print("fwd only", rel(yb, ya), [c[0] for c in calls], [float(c[1].abs().sum()) for c in calls])
pl = list(eng.plans.values())[0]
print("sync points", pl.pb.sync_points, "nfwd", pl.n_fwd)

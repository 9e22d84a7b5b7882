import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from torch import nn
from oracle import synth, torch_ref
from litehandnet_amd import litehourglass as lh
from litehandnet_amd.engine import PlanModule

def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

class Ours(PlanModule):
    def __init__(self, mode):
        super().__init__(); self.mode = mode
        self.a = lh.RepBasicUnit(64, 64, "none"); self.b = lh.RepBasicUnit(64, 64, "none")
    def emit(self, pb, x, out=None):
        y = self.a.emit(pb, x)
        if self.mode == 1: return pb.maxpool(y)
        if self.mode == 2: return pb.ew([pb.maxpool(y), y])
        if self.mode == 3: return self.b.emit(pb, pb.maxpool(y))
        if self.mode == 4: return pb.ew([self.b.emit(pb, pb.maxpool(y)), y])
        if self.mode == 5: return pb.ew([self.b.emit(pb, y), y])
        if self.mode == 6: return self.b.emit(pb, y)
        if self.mode == 7: return self.b.emit(pb, pb.ew([y]))
        if self.mode == 8: return pb.ew([y, pb.avgpool(y, y.H, y.W)])
class Ref(nn.Module):
    def __init__(self, mode):
        super().__init__(); self.mode = mode
        self.a = torch_ref.RepBasicUnit(64, 64, "none"); self.b = torch_ref.RepBasicUnit(64, 64, "none")
    def forward(self, x):
        y = self.a(x)
        if self.mode == 1: return F.max_pool2d(y, 2, 2)
        if self.mode == 2: return F.interpolate(F.max_pool2d(y, 2, 2), size=y.shape[2:]) + y
        if self.mode == 3: return self.b(F.max_pool2d(y, 2, 2))
        if self.mode == 4: return F.interpolate(self.b(F.max_pool2d(y, 2, 2)), size=y.shape[2:]) + y
        if self.mode == 5: return self.b(y) + y
        if self.mode == 6: return self.b(y)
        if self.mode == 7: return self.b(y)
        if self.mode == 8: return y + y
x = torch.randn(4, 64, 16, 16, generator=torch.Generator().manual_seed(0))
for mode in range(1, 9):
    o, r = Ours(mode), Ref(mode)
    sd = synth.synth_state_dict(r, 0); r.load_state_dict(sd); o.load_state_dict(sd); o.cuda(); o.train(); r.train()
    xr = x.clone().requires_grad_(); yr = r(xr); g = torch.randn(yr.shape, generator=torch.Generator().manual_seed(1)); yr.backward(g)
    xg = x.clone().cuda().requires_grad_(); yg = o(xg); yg.backward(g.cuda())
    rp = {k: v for k, v in r.named_parameters() if v.grad is not None}
    worst = max((float((p.grad.cpu().double() - rp[k].grad.double()).norm() / (rp[k].grad.double().norm() + 1e-3 * max(float(v.grad.norm()) for v in rp.values()))), k) for k, p in o.named_parameters() if k in rp)
    print("mode", mode, "fwd %.2e dx %.2e worst %.2e %s" % (rel(yg, yr), rel(xg.grad, xr.grad), worst[0], worst[1]))

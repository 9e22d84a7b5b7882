import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from litehandnet_amd.engine import PlanModule

def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

class Mp(PlanModule):
    def emit(self, pb, x, out=None): return pb.maxpool(x)
class Av(PlanModule):
    def emit(self, pb, x, out=None): return pb.avgpool(x, x.H // 2, x.W // 2)
class Up(PlanModule):
    def emit(self, pb, x, out=None): return pb.ew([pb.maxpool(x), x])
class Short(PlanModule):
    def emit(self, pb, x, out=None):
        m = pb.maxpool(x)
        return pb.ew([m, pb.avgpool(x, m.H, m.W)])
class UpDown(PlanModule):
    def emit(self, pb, x, out=None):
        m = pb.maxpool(x)
        s = pb.ew([m, pb.avgpool(x, m.H, m.W)])
        return pb.ew([s, x])
refs = {
 "maxpool": (Mp(), lambda x: F.max_pool2d(x, 2, 2)),
 "avgpool": (Av(), lambda x: F.adaptive_avg_pool2d(x, (x.shape[2] // 2, x.shape[3] // 2))),
 "up+add": (Up(), lambda x: F.interpolate(F.max_pool2d(x, 2, 2), size=x.shape[2:]) + x),
 "short": (Short(), lambda x: F.max_pool2d(x, 2, 2) + F.adaptive_avg_pool2d(x, (x.shape[2] // 2, x.shape[3] // 2))),
 "updown": (UpDown(), lambda x: F.interpolate(F.max_pool2d(x, 2, 2) + F.adaptive_avg_pool2d(x, (x.shape[2] // 2, x.shape[3] // 2)), size=x.shape[2:]) + x),
}
x = torch.randn(2, 64, 8, 8, generator=torch.Generator().manual_seed(0))
for name, (m, f) in refs.items():
    xr = x.clone().requires_grad_(); yr = f(xr); g = torch.randn(yr.shape, generator=torch.Generator().manual_seed(1)); yr.backward(g)
    xg = x.clone().cuda().requires_grad_(); yg = m(xg); yg.backward(g.cuda())
    print(name, "fwd", rel(yg, yr), "dx", rel(xg.grad, xr.grad))

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from litehandnet_amd import _lib
from litehandnet_amd._lib import View, GradView
L = _lib.lib(); dev = torch.device("cuda:0")
def view(t, table=None):
    N, H, W, Cs = t.shape; v = View(); v.data = t.data_ptr(); v.table = table.data_ptr() if table is not None else None; v.gate = None
    v.N, v.H, v.W, v.cstride, v.coff, v.C = N, H, W, Cs, 0, Cs; return v
for (Cc, N, H, W, stride) in [(32, 3, 12, 16, 1), (32, 64, 32, 32, 1), (64, 3, 12, 16, 1), (32, 3, 12, 16, 2)]:
    torch.manual_seed(0)
    x = torch.randn(N, H, W, Cc, device=dev); w = torch.randn(Cc, Cc, 3, 3, device=dev)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.randn(N, Ho, Wo, Cc, device=dev); dz = torch.randn(N, Ho, Wo, Cc, device=dev)
    g = GradView(); g.dz = dz.data_ptr(); g.dpool = None; g.coef = None
    dx = torch.zeros_like(x); dw = torch.zeros(1, w.numel(), device=dev)
    vx, vy = view(x), view(y)
    _lib.check(L.lhn_conv_kxk_bwd(C.byref(vx), _lib.ptr(w), C.byref(vy), C.byref(g), _lib.ptr(dx), 0, _lib.ptr(dw), stride, 1, C.c_int64(0), _lib.stream()))
    xr = x.permute(0, 3, 1, 2).contiguous().requires_grad_(); wr = w.clone().requires_grad_()
    yr = torch.nn.functional.conv2d(xr, wr, stride=stride, padding=1)
    yr.backward(dz.permute(0, 3, 1, 2).contiguous())
    ew = (dw.view_as(w) - wr.grad).abs().amax(dim=(0, 1)) / wr.grad.abs().max()
    ex = (dx.permute(0, 3, 1, 2) - xr.grad).abs().max() / xr.grad.abs().max()
    print(Cc, N, H, W, stride, "dx err %.2e" % ex.item(), "dw err per tap", [f"{v:.1e}" for v in ew.flatten().tolist()])

"""Micro-benchmarks of single liblhn kernels through the C ABI (torch only allocates memory and times)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litehandnet_amd import _lib
from litehandnet_amd._lib import View, GradView
L = _lib.lib()
dev = torch.device("cuda:0")

def view(t, coff=0, Cc=None, table=None, gate=None):
    N, H, W, Cs = t.shape
    v = View(); v.data = t.data_ptr(); v.table = table.data_ptr() if table is not None else None
    v.gate = gate.data_ptr() if gate is not None else None
    v.N, v.H, v.W, v.cstride, v.coff, v.C = N, H, W, Cs, coff, Cc or Cs
    return v

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us

def table(Cs):
    t = torch.empty(3, Cs, device=dev); t[0] = 1.3; t[1] = 0.1; t[2] = 0.0
    return t

st = _lib.stream()
ONLY = os.environ.get("ONLY", "")
print("kernel, shape, us, GB/s(alg)")
for (N, H, Cc) in ([] if ONLY == "pw" else [(64, 128, 32), (64, 64, 64), (64, 32, 64), (64, 8, 64)]):
    x = torch.randn(N, H, H, Cc, device=dev); y = torch.empty_like(x); w = torch.randn(Cc, 1, 3, 3, device=dev)
    stats = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev); tb = table(Cc)
    vx, vy = view(x, table=tb), view(y)
    byts = 2 * x.numel() * 4
    for name, sp in (("dw_fwd+stats", stats.data_ptr()), ("dw_fwd nostats", None)):
        us = timeit(lambda: _lib.check(L.lhn_conv_dw_fwd(C.byref(vx), _lib.ptr(w), C.byref(vy), C.c_void_p(sp), 3, 1, 1, 1, None, st)))
        print(f"{name}, {N}x{H}x{H}x{Cc}, {us:.1f}, {byts / us / 1e3:.0f}")
    # bn bwd reduce
    dz = torch.randn_like(x); save = torch.randn(2 * Cc, device=dev).abs() + 0.5; sums = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev)
    g = GradView(); g.dz = dz.data_ptr(); g.dpool = None; g.coef = None
    us = timeit(lambda: _lib.check(L.lhn_bn_bwd_reduce(C.byref(vx), C.byref(g), _lib.ptr(save), _lib.ptr(sums), None, st)))
    print(f"bn_bwd_reduce, {N}x{H}x{H}x{Cc}, {us:.1f}, {byts / us / 1e3:.0f}")
    coef = torch.randn(3, Cc, device=dev); g.coef = coef.data_ptr()
    dx = torch.empty_like(x); dw = torch.zeros(16, w.numel(), device=dev)
    vyy = view(y, table=tb)
    us = timeit(lambda: _lib.check(L.lhn_conv_dw_bwd(C.byref(vx), _lib.ptr(w), C.byref(vyy), C.byref(g), _lib.ptr(dx), 0, _lib.ptr(dw), 3, 1, 1, 1, 16, C.c_int64(w.numel()), st)))
    print(f"dw_bwd(data+weight), {N}x{H}x{H}x{Cc}, {us:.1f}, {2.5 * byts / us / 1e3:.0f}")
    # copy baseline
    us = timeit(lambda: y.copy_(x))
    print(f"torch copy, {N}x{H}x{H}x{Cc}, {us:.1f}, {byts / us / 1e3:.0f}")
for (N, H, Ci, Co) in [(64, 64, 64, 64), (64, 64, 128, 128), (64, 128, 32, 32), (64, 64, 64, 128)]:
    x = torch.randn(N, H, H, Ci, device=dev); y = torch.empty(N, H, H, Co, device=dev); w = torch.randn(Co, Ci, device=dev)
    stats = torch.zeros(32 * 2 * Co, dtype=torch.float64, device=dev); tb = table(Ci)
    vx, vy = view(x, table=tb), view(y)
    byts = (x.numel() + y.numel()) * 4
    us = timeit(lambda: _lib.check(L.lhn_conv_pw_fwd(C.byref(vx), _lib.ptr(w), None, C.byref(vy), _lib.ptr(stats), 1, None, None, st)))
    print(f"pw_fwd, {N}x{H}x{H} {Ci}->{Co}, {us:.1f}, {byts / us / 1e3:.0f} GB/s, {2 * N * H * H * Ci * Co / us / 1e6:.1f} TFLOP/s")
    dz = torch.randn_like(y); coef = torch.randn(3, Co, device=dev); tby = table(Co)
    g = GradView(); g.dz = dz.data_ptr(); g.dpool = None; g.coef = coef.data_ptr()
    vyy = view(y, table=tby); dx = torch.empty_like(x); dw = torch.zeros(16, w.numel(), device=dev)
    us = timeit(lambda: _lib.check(L.lhn_conv_pw_bwd(C.byref(vx), _lib.ptr(w), C.byref(vyy), C.byref(g), _lib.ptr(dx), 0, _lib.ptr(dw), None, 1, None, 16, C.c_int64(w.numel()), st)))
    print(f"pw_bwd, {N}x{H}x{H} {Ci}->{Co}, {us:.1f}, {(2 * x.numel() + 2 * y.numel()) * 4 / us / 1e3:.0f} GB/s, {4 * N * H * H * Ci * Co / us / 1e6:.1f} TFLOP/s")

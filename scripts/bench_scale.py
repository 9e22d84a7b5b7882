"""Does a kernel's rate depend on the problem size?  pw 64->64, dw 3x3 64ch, pw bwd, dw bwd at batch 64 / 128 / 256 (64x64 maps),
rotating buffer pairs so that every launch reads cold data; plus torch's copy of the same bytes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litehandnet_amd import _lib
from litehandnet_amd._lib import View, GradView
L = _lib.lib(); dev = torch.device("cuda:0"); st = _lib.stream()

def view(t, table=None):
    v = View(); v.data = t.data_ptr(); v.table = table.data_ptr() if table is not None else None; v.gate = None
    v.N, v.H, v.W, v.cstride, v.coff, v.C = t.shape[0], t.shape[1], t.shape[2], t.shape[3], 0, t.shape[3]
    return v

def timeit(fn, reps):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for N in (64, 128, 256):
    NP = max(2, 1200 // N * 64 // 64 // 4)        # pairs: keep ~1 GB between re-uses
    NP = max(2, int(1.1e9 / (2 * N * 64 * 64 * 64 * 4)))
    xs = [torch.randn(N, 64, 64, 64, device=dev) for _ in range(NP)]
    ys = [torch.empty_like(xs[0]) for _ in range(NP)]
    tb = torch.ones(3, 64, device=dev); tb[1] = 0.1; tb[2] = 0.0
    stats = torch.zeros(32 * 2 * 64, dtype=torch.float64, device=dev)
    w1 = torch.randn(64, 64, device=dev); w3 = torch.randn(64, 1, 3, 3, device=dev)
    vx, vy = [view(x, tb) for x in xs], [view(y) for y in ys]
    byts = 2 * xs[0].numel() * 4
    turn = [0]
    def nxt():
        turn[0] = (turn[0] + 1) % NP
        return turn[0]
    def pw():
        i = nxt(); L.lhn_conv_pw_fwd(C.byref(vx[i]), _lib.ptr(w1), None, C.byref(vy[i]), _lib.ptr(stats), 1, None, None, st)
    def dw():
        i = nxt(); L.lhn_conv_dw_fwd(C.byref(vx[i]), _lib.ptr(w3), C.byref(vy[i]), _lib.ptr(stats), 3, 1, 1, 1, None, st)
    def pw0():
        i = nxt(); L.lhn_conv_pw_fwd(C.byref(vx[i]), _lib.ptr(w1), None, C.byref(vy[i]), None, 1, None, None, st)
    def dw0():
        i = nxt(); L.lhn_conv_dw_fwd(C.byref(vx[i]), _lib.ptr(w3), C.byref(vy[i]), None, 3, 1, 1, 1, None, st)
    def cp():
        i = nxt(); ys[i].copy_(xs[i])
    for name, fn in (("pw64->64", pw), ("pw nostat", pw0), ("dw3x3", dw), ("dw nostat", dw0), ("copy", cp)):
        us = timeit(fn, 4 * NP)
        print(f"N={N:4d} {name:10s} {us:8.1f} us  {byts / us / 1e6:6.2f} TB/s", flush=True)
    del xs, ys
    torch.cuda.empty_cache()

"""Direct C-ABI checks of the round-3 backward fusions: lhn_maxpool2_bwd3 with gradient addends against the three separate
calls it replaces, lhn_ew_bwd_multi against one lhn_ew_bwd3 call per source (both through ctypes on the GPU; the whole-model
parity tests exercise them inside plans)."""
import ctypes as C

import numpy as np
import pytest
import torch

from litehandnet_amd import _lib
from litehandnet_amd._lib import View

pytestmark = pytest.mark.gpu


class GradAdds(C.Structure):      # lhn_grad_adds (include/lhn.h)
    _fields_ = [("same", C.c_void_p), ("same_cstride", C.c_int32), ("same_coff", C.c_int32), ("pooled", C.c_void_p),
                ("OH", C.c_int32), ("OW", C.c_int32), ("pooled_cstride", C.c_int32), ("pooled_coff", C.c_int32)]


def _view(t, coff=0, c=None, table=None):
    v = View()
    v.data, v.table, v.gate, v.pend = t.data_ptr(), (table.data_ptr() if table is not None else None), None, None
    v.N, v.H, v.W, v.cstride, v.coff, v.C = t.shape[0], t.shape[1], t.shape[2], t.shape[3], coff, (t.shape[3] - coff if c is None else c)
    return v


def _rand(shape, seed, dev):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32)).to(dev)


@pytest.mark.parametrize("n,h,w,cbuf,coff,c", [(3, 16, 24, 64, 0, 64), (2, 64, 64, 128, 64, 64)])
def test_maxpool_backward_with_gradient_addends(dev, n, h, w, cbuf, coff, c):
    L, st = _lib.lib(), _lib.stream()
    x = _rand((n, h, w, cbuf), 1, dev)
    table = torch.stack([torch.full((cbuf,), 0.7), torch.full((cbuf,), -0.1), torch.full((cbuf,), 0.25)]).to(dev).contiguous()   # scale | shift | slope
    y = torch.empty(n, h // 2, w // 2, cbuf, device=dev)
    vx, vy = _view(x, coff, c, table), _view(y, coff, c)
    _lib.check(L.lhn_maxpool2_fwd(C.byref(vx), C.byref(vy), st), "maxpool fwd")
    dy = _rand((n, h // 2, w // 2, cbuf), 2, dev)
    same = _rand((n, h, w, 96), 3, dev)                 # gradient of a sum that reads x: channels [16, 16 + c) of a 96-channel buffer
    pooled = _rand((n, 8, 8, cbuf), 4, dev)             # gradient of adaptive_avg_pool2d(x, 8)
    # reference: three read-modify-write passes
    dx0 = torch.zeros_like(x)
    vs = _view(same, 16, c)
    _lib.check(L.lhn_ew_bwd3(C.byref(vx), C.byref(vs), _lib.ptr(same), None, C.c_float(1.0), _lib.ptr(dx0), 0, None, st), "ew bwd")
    _lib.check(L.lhn_avgpool_bwd3(C.byref(vx), _lib.ptr(pooled), 8, 8, cbuf, coff, _lib.ptr(dx0), 1, None, st), "avgpool bwd")
    _lib.check(L.lhn_maxpool2_bwd2(C.byref(vx), C.byref(vy), _lib.ptr(dy), _lib.ptr(dx0), 1, None, st), "maxpool bwd")
    # fused
    dx1 = torch.zeros_like(x)
    ad = GradAdds()
    ad.same, ad.same_cstride, ad.same_coff = same.data_ptr(), 96, 16
    ad.pooled, ad.OH, ad.OW, ad.pooled_cstride, ad.pooled_coff = pooled.data_ptr(), 8, 8, cbuf, coff
    _lib.check(L.lhn_maxpool2_bwd3(C.byref(vx), C.byref(vy), _lib.ptr(dy), _lib.ptr(dx1), 0, None, C.byref(ad), st), "maxpool bwd3")
    torch.cuda.synchronize()
    a, b = dx0[..., coff:coff + c], dx1[..., coff:coff + c]
    assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())        # same terms, another order of the three additions
    assert float(dx1[..., :coff].abs().max() if coff else 0.0) == 0.0        # channels outside the view untouched
    # odd sizes are refused (a pixel outside every pooling window would miss its addends)
    xo = _rand((1, 7, 8, c), 5, dev)
    yo = torch.empty(1, 4, 4, c, device=dev)
    rc = L.lhn_maxpool2_bwd3(C.byref(_view(xo)), C.byref(_view(yo)), _lib.ptr(yo), _lib.ptr(torch.zeros_like(xo)), 0, None, C.byref(ad), st)
    assert rc != 0 and b"even" in L.lhn_last_error()


@pytest.mark.parametrize("nsrc,slope", [(2, 0.01), (3, 1.0), (2, 0.0)])
def test_combine_backward_multi_source(dev, nsrc, slope):
    L, st = _lib.lib(), _lib.stream()
    n, h, w, c = 3, 12, 20, 32
    dst = _rand((n, h, w, 64), 11, dev)                 # the combine's output (its sign picks the activation derivative)
    ddst = _rand((n, h, w, 64), 12, dev)
    vd = _view(dst, 32, c)
    srcs = [_rand((n, h, w, c + 8 * k), 20 + k, dev) for k in range(nsrc)]
    prior = [_rand((n, h, w, c + 8 * k), 30 + k, dev) for k in range(nsrc)]
    acc = [k % 2 for k in range(nsrc)]                  # store / accumulate mixed
    ref = [p.clone() for p in prior]
    for k in range(nsrc):
        _lib.check(L.lhn_ew_bwd3(C.byref(_view(srcs[k], 8 * k, c)), C.byref(vd), _lib.ptr(ddst), None, C.c_float(slope), _lib.ptr(ref[k]),
                                 acc[k], None, st), "ew bwd3")
    out = [p.clone() for p in prior]
    views = (View * nsrc)(*[_view(srcs[k], 8 * k, c) for k in range(nsrc)])
    dptr = (C.c_void_p * nsrc)(*[o.data_ptr() for o in out])
    accs = (C.c_int * nsrc)(*acc)
    _lib.check(L.lhn_ew_bwd_multi(views, nsrc, C.byref(vd), _lib.ptr(ddst), None, C.c_float(slope), dptr, accs, None, st), "ew bwd multi")
    torch.cuda.synchronize()
    for k in range(nsrc):
        assert torch.equal(ref[k], out[k]), k           # same arithmetic per element: bit-identical

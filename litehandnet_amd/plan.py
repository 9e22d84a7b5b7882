"""Static execution plan for the litehandnet backbone on one MI355X.

The reference runs ~330 unfused torch ops per forward.  Here the Python
mirror of the module tree (`liteHandNet.py`, `litehourglass.py`, ...) is walked ONCE per input
shape by a `PlanBuilder`; the result is a flat list of kernel calls over one workspace arena that
`lhn_plan_run` (csrc/lhn_plan.cpp) enqueues with a single C call per forward / backward.

Conventions
  * activations: NHWC fp32 buffers in the arena, each with a per-channel table (scale, shift,
    slope) and optionally a per-(n,c) gate; a conv writes its RAW output plus BatchNorm statistics,
    `FINALIZE` turns the statistics into the table, consumers apply it on load (no BN/act pass).
  * cat / chunk / channel slices are (buffer, coff, C) views -- producers write into slices.
  * gradients: one grad buffer per activation buffer holding d(loss)/d(consumed value); the
    backward list is generated here by walking the forward records in reverse and tracking which
    channel ranges have been written (store) or must be accumulated.
"""
import ctypes as C
import os
from dataclasses import dataclass, field

import torch

from . import _lib
from ._lib import Buf, Op

# op kinds (must match csrc/lhn_plan.cpp)
STEM, PW, DW, KXK, FINALIZE, EW, MAXPOOL, AVGPOOL, CA_MLP, TABLE_FILL, MEMSET, ATT_MLP, SE_MLP, SHUFFLE = range(1, 15)
STEM_BWD, PW_BWD, DW_BWD, KXK_BWD, BN_BWD, EW_BWD, MAXPOOL_BWD, AVGPOOL_BWD, GATE_REDUCE, CA_MLP_BWD, ATT_MLP_BWD, SE_MLP_BWD, SHUFFLE_BWD = range(101, 114)
SLOPE_SILU = 2.0       # LHN_SLOPE_SILU in include/lhn.h: the combine applies SiLU instead of a leaky ReLU
SLOPE_RELU_SIGMOID = 3.0   # LHN_SLOPE_RELU_SIGMOID: sigmoid(relu(v)) (lite_hrnet.py: nn.ReLU followed by nn.Sigmoid)
EW_MUL, EW_BILINEAR = 1, 2   # EwSrcs.mode bits: product of the sources / bilinear (align_corners) resampling of smaller ones

ALIGN = 256
STAT_REPLICAS = 32     # LHN_STAT_REPLICAS in include/lhn.h
TICKET_WORDS = 33      # arrival counters of a fused finalize (lhn_bnfin.counter: one top word + 32 group words)


def _al(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


@dataclass
class TRef:
    """A (buffer, channel-slice) view.  buf == -1 is the NCHW input image."""
    buf: int
    coff: int
    C: int
    H: int
    W: int


@dataclass
class TCat:
    """Channel concatenation of views that live in DIFFERENT buffers (same N, H, W) -- the result of a torch.cat whose
    operands are never copied together: RepBasicUnit's pass-through half stays where it is (litehourglass.py:74-77) and
    every whole-tensor consumer (max-pool, residual add, average pool) runs once per part."""
    parts: list

    @property
    def C(self):
        return sum(p.C for p in self.parts)

    @property
    def H(self):
        return self.parts[0].H

    @property
    def W(self):
        return self.parts[0].W


def _parts(x):
    return list(x.parts) if isinstance(x, TCat) else [x]


@dataclass
class _BufRec:
    H: int
    W: int
    C: int
    gate: bool = False
    coef: bool = False
    dpool: bool = False
    lazy: object = None      # the EW record of a sum that is taken ON LOAD by its consumers (never written in forward)
    off: dict = field(default_factory=dict)


class _Arena:
    def __init__(self):
        self.size = 0

    def take(self, nbytes):
        off = self.size
        self.size += _al(nbytes)
        return off


class _IdentConv:
    weight = None
    bias = None


_IDENT = _IdentConv()


class PlanBuilder:
    def __init__(self, N, state_index, image_hw=None, with_backward=True, p_drop=0.0):
        self.N = N
        self.state_index = state_index      # id(tensor) -> index in the params array
        self.bufs = []
        self.recs = []                      # forward records (python dicts)
        self.with_backward = with_backward
        self.p_drop = p_drop
        self.image_hw = image_hw
        # arenas: zf = zeroed at the start of every forward, zb = zeroed at the start of every backward
        self.ar = {"zf": _Arena(), "zb": _Arena(), "mask": _Arena(), "misc": _Arena()}
        self.out_ref = None
        self.nchw_out_C = None
        self.nchw_stacks = 1
        self.in_ref = None

    # ------------------------------------------------------------------ declarations
    def image(self):
        H, W = self.image_hw
        return TRef(-1, 0, 3, H, W)

    def buffer(self, H, W, C):
        self.bufs.append(_BufRec(H, W, C))
        return len(self.bufs) - 1

    def new(self, H, W, C):
        return TRef(self.buffer(H, W, C), 0, C, H, W)

    def input_tensor(self, C, H, W):
        self.in_ref = self.new(H, W, C)
        return self.in_ref

    def slice(self, x, coff, C):
        if not (0 <= coff and coff + C <= x.C and coff % 4 == 0 and C % 4 == 0):
            raise _lib.LhnError(f"channel slice [{coff}:{coff + C}] of {x.C} channels: slices start and end on multiples of 4 "
                                "(the kernels move 4 channels per thread)")
        if isinstance(x, TCat):
            out, lo = [], 0
            for p in x.parts:
                a, b = max(coff, lo), min(coff + C, lo + p.C)
                if a < b:
                    out.append(TRef(p.buf, p.coff + a - lo, b - a, p.H, p.W))
                lo += p.C
            return out[0] if len(out) == 1 else TCat(out)
        return TRef(x.buf, x.coff + coff, C, x.H, x.W)

    def cat(self, xs):
        """torch.cat(xs, dim=1) without a copy."""
        parts = [p for x in xs for p in _parts(x)]
        assert all((p.H, p.W) == (parts[0].H, parts[0].W) for p in parts)
        return parts[0] if len(parts) == 1 else TCat(parts)

    def single(self, x):
        """A one-buffer view of x (convolutions and gates read ONE buffer): multi-part tensors are copied together."""
        return self.ew([x]) if isinstance(x, TCat) else x

    def owns_buffer(self, x):
        """True when x is the whole of an ungated buffer that is not the block's input (a gate attaches to a buffer)."""
        if isinstance(x, TCat):
            return False
        self.real(x)
        b = self.bufs[x.buf]
        return x.coff == 0 and x.C == b.C and not b.gate and not (self.in_ref is not None and x.buf == self.in_ref.buf)

    @staticmethod
    def _segments(tensors):
        """Channel ranges (lo, hi) on which every tensor of the list is ONE part."""
        cuts = set()
        for t in tensors:
            lo = 0
            for p in _parts(t):
                cuts.add(lo)
                lo += p.C
            cuts.add(lo)
        cuts = sorted(cuts)
        return list(zip(cuts[:-1], cuts[1:]))

    def _ws(self, arena, nbytes):
        return (arena, self.ar[arena].take(nbytes))

    def _p(self, t):
        return -1 if t is None else self.state_index[id(t)]

    # ------------------------------------------------------------------ forward emitters
    def conv(self, x, conv, bn=None, slope=1.0, out=None, nchw_out=False, stack=None, bn_repeat=1):
        """conv (+ train/eval BatchNorm + leaky slope as a pending transform).  Returns the output view.
        1x1 convolutions take any channel counts: views are padded to multiples of 4 (a 21-feature head is a 24-channel
        NHWC buffer whose last channels are exact zeros) and the library slices wide ones.  `stack = (i, S)`: the NCHW output
        is slot i of an [N, S, K, H, W] tensor (hourglassnet.py:136)."""
        x = self.single(x)
        cout, cin_g, kh, kw = conv.weight.shape
        s, p, d, g = conv.stride[0], conv.padding[0], conv.dilation[0], conv.groups
        Ho = (x.H + 2 * p - d * (kh - 1) - 1) // s + 1
        Wo = (x.W + 2 * p - d * (kw - 1) - 1) // s + 1
        if x.buf == -1:
            kind = STEM
        elif g == cout and cin_g == 1 and x.C == cout:
            kind = DW
        elif kh == 1 and g == 1:
            kind = PW
            assert p == 0
        elif kh == 3 and g == 1 and p == 1 and d == 1:
            kind = KXK
        else:
            raise _lib.LhnError(f"unsupported convolution {tuple(conv.weight.shape)} groups={g}")
        cpad = cout
        if kind == PW:
            assert cin_g <= x.C < cin_g + 4, (cin_g, x.C)
            if not nchw_out:
                cpad = (cout + 3) // 4 * 4          # (a BatchNorm behind a padded output owns the first `cout` columns)
        elif kind == KXK:
            assert cin_g == x.C, (cin_g, x.C)
        xs = self.lazy_sources(x)
        if xs is not None:
            ok = (kind == DW and len(xs) <= 2 and kh == 3 and s == 1 and p == d and d in (1, 2) and x.C % 32 == 0 and x.W >= (8 if d == 1 else 16)
                  and conv.weight is not None) or \
                 (kind == PW and s == 1 and not nchw_out and x.C == cout and x.C in self.LAZY_PW and cin_g == x.C)
            if not ok:
                self.real(x)
                xs = None
        if nchw_out:
            assert kind == PW and bn is None and out is None
            out = TRef(-2, 0, cout, Ho, Wo)
            self.nchw_out_C = cout
            self.nchw_stacks = 1 if stack is None else int(stack[1])
        elif out is None:
            out = self.new(Ho, Wo, cpad)
        assert (out.H, out.W, out.C) == (Ho, Wo, cpad), (out, Ho, Wo, cpad)
        if out.buf >= 0:
            self.real(out)
        rec = dict(op=kind, x=x, out=out, conv=conv, bn=bn, slope=float(slope), k=kh, stride=s, pad=p, dil=d,
                   nchw=nchw_out, stack=(0, 1) if stack is None else (int(stack[0]), int(stack[1])),
                   wrc=(cout if cpad != cout else 0, cin_g if cin_g != x.C else 0), xs=xs, bn_repeat=int(bn_repeat))
        if kind == KXK:
            rec["wt"] = self._ws("misc", 9 * cout * cin_g * 4)      # tap-major weight scratch (lhn_conv_kxk_*: wt_scratch)
        if bn is not None:
            sc = max(cout, cpad)        # statistics are laid out for the (padded) output view
            rec["stats"] = self._ws("zf", STAT_REPLICAS * 2 * sc * 8)
            rec["cnt"] = self._ws("zf", 4 * TICKET_WORDS)
            rec["save"] = self._ws("misc", 2 * sc * 4)
            if self.with_backward:
                rec["sums"] = self._ws("zb", STAT_REPLICAS * 2 * sc * 8)
                rec["bcnt"] = self._ws("zb", 4 * TICKET_WORDS)
                self.bufs[out.buf].coef = True
        self.recs.append(rec)
        if bn is None and not nchw_out:
            # BN-free conv (deployed RepConv/RepBlock, repblocks.py:41-43,118-119): bias and activation stay pending
            # in the output's table.  The 1x1 kernel already adds its bias while storing.
            tb = conv.bias if kind != PW else None
            if tb is not None or float(slope) != 1.0:
                if self.with_backward:
                    raise _lib.LhnError("BN-free convolutions with a pending bias/activation are inference-only (deploy form)")
                self.recs.append(dict(op=TABLE_FILL, out=out, bias=tb, slope=float(slope)))
        return out

    def bn_only(self, x, bn, slope=1.0):
        """BatchNorm applied straight to a tensor (RepBlock.rbr_identity, repblocks.py:113-114): lowered as an
        identity depthwise 1x1 (weights = NULL = ones), which copies the consumed value, takes the batch
        statistics in its epilogue and leaves the normalisation pending like any other conv+BN."""
        x = self.real(self.single(x))
        out = self.new(x.H, x.W, x.C)
        rec = dict(op=DW, x=x, out=out, conv=_IDENT, bn=bn, slope=float(slope), k=1, stride=1, pad=0, dil=1, nchw=False)
        rec["stats"] = self._ws("zf", STAT_REPLICAS * 2 * x.C * 8)
        rec["cnt"] = self._ws("zf", 4 * TICKET_WORDS)
        rec["save"] = self._ws("misc", 2 * x.C * 4)
        if self.with_backward:
            rec["sums"] = self._ws("zb", STAT_REPLICAS * 2 * x.C * 8)
            rec["bcnt"] = self._ws("zb", 4 * TICKET_WORDS)
            self.bufs[out.buf].coef = True
        self.recs.append(rec)
        return out

    # Residual sums whose readers can add the operands while loading them (the 3x3 depthwise LDS kernel: 2 operands, the
    # square 64/128-channel 1x1: 3) are "lazy": the combine is recorded but not launched in forward; the backward pass
    # materialises it once (weight gradients need the value) and distributes its gradient like any other combine.
    LAZY_PW = (64, 128)

    def lazy_sources(self, x):
        """[(TRef, coef)] behind a lazy view x (same channel slice of every operand), or None."""
        if isinstance(x, TCat) or x.buf < 0:
            return None
        rec = self.bufs[x.buf].lazy
        if rec is None:
            return None
        return [(TRef(t.buf, t.coff + x.coff, x.C, t.H, t.W), c) for t, c in rec["flat"]]

    def real(self, x):
        """x as something every kernel can read: a lazy sum is materialised where it was recorded."""
        for p in _parts(x):
            if p.buf >= 0 and self.bufs[p.buf].lazy is not None:
                self.bufs[p.buf].lazy["lazy"] = False
                self.bufs[p.buf].lazy = None
        return x

    def ew(self, srcs, out_slope=1.0, out=None, lazy=False, mode=0, coefs=None):
        """out = act(sum of sources); smaller sources are nearest-upsampled.  Plain output.
        lazy=True (plain same-size sum into a fresh buffer): see lazy_sources().
        mode EW_MUL: product instead of sum (exactly two single-buffer sources); EW_BILINEAR: smaller sources are resampled
        bilinearly with align_corners=True.  coefs: one factor per source."""
        assert 1 <= len(srcs) <= 3
        if mode or coefs is not None:
            srcs = [self.real(self.single(t)) for t in srcs]
            assert not (mode & EW_MUL) or len(srcs) == 2
            H, W = max(t.H for t in srcs), max(t.W for t in srcs)
            if out is None:
                out = self.new(H, W, srcs[0].C)
            assert not isinstance(out, TCat) and all(t.C == out.C for t in srcs)
            self.recs.append(dict(op=EW, srcs=list(srcs), out=self.real(out), slope=float(out_slope), mode=int(mode),
                                  coefs=None if coefs is None else [float(c) for c in coefs]))
            return out
        if lazy and out is None and out_slope == 1.0 and not any(isinstance(t, TCat) for t in srcs) and \
                all((t.H, t.W, t.C) == (srcs[0].H, srcs[0].W, srcs[0].C) and t.buf >= 0 for t in srcs):
            flat = []
            for t in srcs:
                inner = self.lazy_sources(t)
                for u, c in (inner if inner is not None else [(t, 1.0)]):
                    for j, (v, cv) in enumerate(flat):
                        if (v.buf, v.coff, v.C) == (u.buf, u.coff, u.C):
                            flat[j] = (v, cv + c)
                            break
                    else:
                        flat.append((u, c))
            if len(flat) <= 3:
                out = self.new(srcs[0].H, srcs[0].W, srcs[0].C)
                rec = dict(op=EW, srcs=list(srcs), out=out, slope=1.0, lazy=True, flat=flat)
                self.bufs[out.buf].lazy = rec
                self.recs.append(rec)
                return out
        srcs = [self.real(t) for t in srcs]
        if out is not None:
            self.real(out)
        if out_slope == SLOPE_SILU and len(srcs) > 1:
            # the SiLU backward recomputes the pre-activation from its (single, same-size) source: sum first
            return self.ew([self.ew(srcs, 1.0)], SLOPE_SILU, out)
        H, W = max(s.H for s in srcs), max(s.W for s in srcs)
        if out is None:
            out = self.new(H, W, srcs[0].C)
        assert all(s.C == out.C for s in srcs)
        if any(isinstance(t, TCat) for t in list(srcs) + [out]):
            for lo, hi in self._segments(list(srcs) + [out]):      # one launch per run of channels that is one part everywhere
                self.ew([self.slice(t, lo, hi - lo) for t in srcs], out_slope, self.slice(out, lo, hi - lo))
            return out
        self.recs.append(dict(op=EW, srcs=list(srcs), out=out, slope=float(out_slope)))
        return out

    def maxpool(self, x, out=None):
        x = self.real(x)
        Ho, Wo = (x.H + 1) // 2, (x.W + 1) // 2
        if out is None:
            out = self.new(Ho, Wo, x.C)
        if isinstance(x, TCat) or isinstance(out, TCat):
            for lo, hi in self._segments([x, out]):
                self.maxpool(self.slice(x, lo, hi - lo), self.slice(out, lo, hi - lo))
            return out
        self.recs.append(dict(op=MAXPOOL, x=x, out=out))
        return out

    def avgpool(self, x, OH, OW, out=None):
        """adaptive_avg_pool2d of the consumed value -> a plain [N,OH,OW,C] buffer (one per part of a multi-part x), or into the
        channel slice `out` of an existing one (the pooled branches of CrossResolutionWeighting land side by side)."""
        x = self.real(x)
        if isinstance(x, TCat):
            assert out is None
            return TCat([self.avgpool(p, OH, OW) for p in x.parts])
        if out is None:
            out = self.new(OH, OW, x.C)
        assert (out.H, out.W, out.C) == (OH, OW, x.C)
        self.recs.append(dict(op=AVGPOOL, x=x, out=out, OH=OH, OW=OW, ca=False))
        return out

    def shuffle2(self, a, b):
        """channel_shuffle(torch.cat([a, b], 1), 2) (lite_hrnet.py:29-52): a new plain buffer with a / b interleaved."""
        a, b = self.real(self.single(a)), self.real(self.single(b))
        assert (a.H, a.W, a.C) == (b.H, b.W, b.C) and a.C % 2 == 0 and a.coff % 2 == 0 and b.coff % 2 == 0
        out = self.new(a.H, a.W, 2 * a.C)
        self.recs.append(dict(op=SHUFFLE, a=a, b=b, out=out))
        return out

    def channel_attention(self, y, ca):
        """common.py:40-66 on the WHOLE buffer behind `y`: sets the buffer's gate.  Returns y."""
        b = self.bufs[y.buf]
        assert y.coff == 0 and y.C == b.C, "channel attention gates a whole buffer"
        assert not b.gate, "buffer is already gated"
        Cc = y.C
        pooled = self._ws("misc", self.N * 9 * Cc * 4)
        save = self._ws("misc", (6 * self.N * Cc + 2 * Cc) * 4)          # forward rows + backward scratch (lhn_ca_mlp_bwd)
        mask = self._ws("mask", self.N * Cc * 4) if self.p_drop > 0 else None
        rec = dict(op=CA_MLP, y=y, ca=ca, pooled=pooled, save=save, mask=mask, gsum=self._ws("misc", 2 * Cc * 8))
        # gated RepBasicUnit: the copy of the pass-through half into this buffer rides in the attention's pooling launch
        # (lhn_avgpool_fwd4: copy_src); the record stays for the backward pass.  LHN_COPY_POOL=0: separate copy.
        copies = [q for q in self.recs if q["op"] == EW and not q.get("lazy") and "flat" not in q and not q.get("mode") and
                  q.get("coefs") is None and not isinstance(q["out"], TCat) and q["out"].buf == y.buf]
        writers = [q for q in self.recs if q["op"] in (STEM, PW, DW, KXK, MAXPOOL, AVGPOOL, SHUFFLE) and
                   q.get("out") is not None and not isinstance(q["out"], TCat) and q["out"].buf == y.buf]
        if os.environ.get("LHN_COPY_POOL", "1") != "0" and len(copies) == 1 and not hasattr(ca, "rbr_reparam"):
            q = copies[0]
            src = q["srcs"][0] if len(q["srcs"]) == 1 else None
            if src is not None and not isinstance(src, TCat) and src.buf >= 0 and src.buf != y.buf and q["slope"] == 1.0 and \
                    q["out"].coff == 0 and 0 < q["out"].C < Cc and (src.H, src.W) == (y.H, y.W) and \
                    all(w["out"].coff >= q["out"].C for w in writers) and self.bufs[src.buf].lazy is None:
                q["fwd_fused"] = True
                rec["copy"] = q
        if self.with_backward:
            rec["gsum_b"] = self._ws("misc", 2 * Cc * 8)
            rec["dgate"] = self._ws("zb", 3 * self.N * Cc * 4)         # dgate | T0, T1 of the gate-gradient pass (bnslices); zeroed with the arena
            b.dpool = True
            # BatchNorm-backward sums of the convolutions that wrote this buffer: assembled by the attention's backward from the
            # gate-gradient pass and the forward pooling pass (include/lhn.h: lhn_bn_slices) -- their lhn_bn_bwd_reduce passes
            # over the feature map and its gradient disappear.  LHN_GATE_BN_SUMS=0: separate passes.
            prods = [q for q in self.recs if q["op"] in (STEM, PW, DW, KXK) and q["bn"] is not None and
                     not isinstance(q["out"], TCat) and q["out"].buf == y.buf]
            if os.environ.get("LHN_GATE_BN_SUMS", "1") != "0" and 1 <= len(prods) <= 2 and not hasattr(ca, "rbr_reparam") and \
                    all(not q.get("wrc", (0, 0))[0] and q.get("bn_repeat", 1) == 1 for q in prods):
                rec["bnslices"] = prods
                rec["pstat"] = self._ws("misc", self.N * 9 * 2 * Cc * 4)
                for q in prods:
                    q["sums_by_ca"] = True
        self.recs.append(rec)
        b.gate = True
        return y

    def me_attention(self, y, att):
        """`mynet` attention (pose_hg_ms_att.py:165-174) on the WHOLE plain buffer behind `y`; `att` is the reference's
        nn.Sequential (1 = BatchNorm2d, 3 = depthwise 3x3 conv, 6 = Linear).  Sets the buffer's gate, returns y."""
        b = self.bufs[y.buf]
        assert y.coff == 0 and y.C == b.C and not b.gate, "attention gates a whole, ungated buffer"
        Cc = y.C
        rec = dict(op=ATT_MLP, y=y, att=att, pooled=self._ws("misc", self.N * 9 * Cc * 4),
                   save=self._ws("misc", (3 * self.N * Cc + 2 * Cc) * 4),
                   mask=self._ws("mask", self.N * Cc * 4) if self.p_drop > 0 else None, gsum=self._ws("misc", 2 * Cc * 8))
        if self.with_backward:
            rec["gsum_b"] = self._ws("misc", 2 * Cc * 8)
            rec["dgate"] = self._ws("zb", self.N * Cc * 4)
            b.dpool = True
        self.recs.append(rec)
        b.gate = True
        return y

    def se_attention(self, y, se, convs=None, mode=0):
        """SEBlock (common.py:23-37) on the WHOLE buffer behind `y` (square maps: avg_pool2d(kernel=W) is global).
        convs = (down, up) Conv2d modules when they are not `se.down` / `se.up`; mode 1 = SpatialWeighting of
        lite_hrnet.py:55-74 (global average pool, sigmoid(relu(.)) after both convolutions)."""
        b = self.bufs[y.buf]
        assert y.coff == 0 and y.C == b.C and not b.gate, "attention gates a whole, ungated buffer"
        if y.H != y.W and mode == 0:
            raise _lib.LhnError("SEBlock pools with kernel_size = width: only square maps are built")
        down, up = convs if convs is not None else (se.down, se.up)
        Cc, J = y.C, down.weight.shape[0]
        rec = dict(op=SE_MLP, y=y, se=se, down=down, up=up, mode=int(mode), J=J, pooled=self._ws("misc", self.N * Cc * 4),
                   save=self._ws("misc", self.N * (J + Cc) * 4))
        if self.with_backward:
            rec["dgate"] = self._ws("zb", self.N * Cc * 4)
            b.dpool = True
        self.recs.append(rec)
        b.gate = True
        return y

    def set_output(self, y):
        """Generic (block-level) output: materialise the consumed value into a plain buffer."""
        self.out_ref = self.ew([y])
        return self.out_ref

    # ------------------------------------------------------------------ lowering
    def _layout(self):
        N = self.N
        base = 0
        self.arena_base = {}
        for name in ("zf", "zb", "mask", "misc"):
            self.arena_base[name] = base
            base += _al(self.ar[name].size)
        self.table_base = base
        for b in self.bufs:
            b.off["table"] = base
            base += _al(3 * b.C * 4)
        self.table_end = base
        for b in self.bufs:
            if b.gate:
                b.off["gate"] = base
                base += _al(N * b.C * 4)
            if self.with_backward and b.coef:
                b.off["coef"] = base
                base += _al(3 * b.C * 4)
            if self.with_backward and b.dpool:
                b.off["dpool"] = base
                base += _al(N * 25 * b.C * 4)      # LHN_DPOOL_SLOTS: 5 x 5 bin-overlap segments
        for b in self.bufs:
            b.off["data"] = base
            if b.lazy is None or self.with_backward:      # a lazy sum is only ever written by the backward pass
                base += _al(N * b.H * b.W * b.C * 4)
        self.act_bytes = base
        if self.with_backward:
            for b in self.bufs:
                b.off["grad"] = base
                base += _al(N * b.H * b.W * b.C * 4)
        self.total_bytes = base

    def _abs(self, ref):
        return -1 if ref is None else self.arena_base[ref[0]] + ref[1]

    @staticmethod
    def _mk(kind, ins=(), out=None, p=(), ws=(), i=(), f=()):
        o = Op()
        o.kind = kind
        for k in range(3):
            o.in_buf[k] = -1
        for k, t in enumerate(ins):
            o.in_buf[k], o.in_coff[k], o.in_C[k] = t.buf, t.coff, t.C
        if out is not None:
            o.out_buf, o.out_coff, o.out_C = out.buf, out.coff, out.C
        else:
            o.out_buf = -1
        for k in range(12):
            o.p[k] = p[k] if k < len(p) else -1
        for k in range(12):
            o.ws[k] = ws[k] if k < len(ws) else -1
        for k in range(8):
            o.i[k] = i[k] if k < len(i) else 0
        for k in range(8):
            o.f[k] = f[k] if k < len(f) else 0.0
        return o

    def _xs(self, r):
        """(input views, number of sources, coefficient slots f[4..6]) of a convolution record: the operands of a lazy sum
        when the input still is one, else the input itself."""
        x = r["x"]
        if r.get("xs") is not None and x.buf >= 0 and self.bufs[x.buf].lazy is not None:
            views = [t for t, _ in r["xs"]]
            coefs = [c for _, c in r["xs"]]
            return views, len(views), coefs + [0.0] * (3 - len(coefs))
        return [x], 1, [1.0, 0.0, 0.0]

    def _grad_mode(self, written, t):
        """1 = store, 2 = accumulate for a write of d(value) into view t; updates the tracker."""
        ranges = written.setdefault(t.buf, [])
        lo, hi = t.coff, t.coff + t.C
        overlap = [r for r in ranges if r[0] < hi and lo < r[1]]
        if not overlap:
            ranges.append((lo, hi))
            return 1
        covered = sorted(overlap)
        cur = lo
        for a, b in covered:
            if a > cur:
                break
            cur = max(cur, b)
        if cur >= hi:
            return 2
        # partial overlap: fall back to a zeroed gradient buffer with accumulate-only writes
        self._needs_zero_grad.add(t.buf)
        ranges.append((lo, hi))
        return 2

    def _covered(self, written, t):
        """The backward op of t's producer is about to READ d(t): channels nobody wrote (an output only partly consumed) must
        read as zero -> the gradient buffer is zero-filled at the start of the backward and every write accumulates."""
        if t is None or isinstance(t, TCat) or t.buf < 0:
            return
        buf = t.buf
        while buf in self._alias_of:            # an aliased gradient buffer is written through the combine it aliases
            buf = self._alias_of[buf]
        if buf != t.buf:
            return
        cur, hi = t.coff, t.coff + t.C
        for a, b in sorted(written.get(t.buf, [])):
            if a > cur:
                break
            cur = max(cur, b)
        if cur < hi:
            self._needs_zero_grad.add(t.buf)
            written.setdefault(t.buf, []).append((t.coff, hi))

    def finalize(self):
        self._layout()
        N = self.N
        fwd, bwd = [], []
        mk = self._mk
        # ---------------- forward
        if self.ar["zf"].size:
            fwd.append(mk(MEMSET, ws=(self.arena_base["zf"], self.ar["zf"].size)))
        # Plans with a backward: a convolution that sums a lazy residual on load also WRITES the sum (lhn_pw_opts.sum_out) when
        # the readers' channel slices cover the buffer -- the weight gradients of the backward then find it in memory and the
        # re-materialising combine (which re-reads every operand) is not needed.  LHN_SUM_OUT=0: re-materialise.
        self.sum_out = {}               # id(conv record) -> (byte offset of the sum's buffer, pixel stride * 65536 + channel)
        self._sum_written = set()       # lazy buffers complete after the forward
        if self.with_backward and os.environ.get("LHN_SUM_OUT", "1") != "0":
            cover = {}
            for r in self.recs:
                if r["op"] not in (PW, DW) or isinstance(r["x"], TCat):
                    continue
                x = r["x"]
                if self._xs(r)[1] < 2 or (r["op"] == PW and (x.C > 128 or r["out"].C > 128)):
                    continue
                spans = cover.setdefault(x.buf, [])
                if any(a < x.coff + x.C and x.coff < b for a, b, _ in spans):
                    continue
                spans.append((x.coff, x.coff + x.C, r))
            for b, spans in cover.items():
                spans.sort(key=lambda t: t[0])
                C = self.bufs[b].C
                if spans[0][0] != 0 or spans[-1][1] != C or any(p[1] != q[0] for p, q in zip(spans, spans[1:])):
                    continue
                self._sum_written.add(b)
                for lo, hi, r in spans:
                    self.sum_out[id(r)] = (self.bufs[b].off["data"], C * 65536 + lo)
        for r in self.recs:
            k = r["op"]
            if k in (STEM, PW, DW, KXK):
                conv, bn, x, out = r["conv"], r["bn"], r["x"], r["out"]
                stats = self._abs(r.get("stats"))
                so = self.sum_out.get(id(r), (-1, -1))
                pw = self._p(conv.weight)
                # a trailing BatchNorm rides on the conv op: the last workgroup of the conv finalizes the table
                if bn is not None:
                    pbn = (self._p(bn.weight), self._p(bn.bias), self._p(bn.running_mean), self._p(bn.running_var),
                           self._p(bn.num_batches_tracked))
                    wsl = (stats, self._abs(r["save"]), self._abs(r["cnt"]))
                    fl = (bn.eps, bn.momentum, r["slope"], float(r.get("bn_repeat", 1)))
                else:
                    pbn, wsl, fl = (-1, -1, -1, -1, -1), (stats,), ()
                cb = self._p(getattr(conv, "bias", None)) if bn is not None else -1   # biased conv + BN: bias goes to the finalize
                if k == STEM:
                    fwd.append(mk(STEM, out=out, p=(pw, cb) + pbn, ws=wsl, i=(r["k"], r["stride"], r["pad"], x.H, x.W), f=fl))
                elif k == PW:
                    o = TRef(-1, 0, out.C, out.H, out.W) if r["nchw"] else out
                    xv, nx, cf = self._xs(r)
                    fwd.append(mk(PW, ins=xv, out=o, p=(pw, self._p(conv.bias)) + pbn, ws=(tuple(wsl) + (-1,) * 4)[:4] + so,
                                  i=(r["stride"], 1 if r["nchw"] else 0, r["wrc"][0], r["wrc"][1], r["stack"][0], r["stack"][1], nx),
                                  f=(tuple(fl) + (0.0,) * 4)[:4] + tuple(cf)))
                elif k == DW:
                    xv, nx, cf = self._xs(r)
                    fwd.append(mk(DW, ins=xv, out=out, p=(pw, cb) + pbn, ws=(tuple(wsl) + (-1,) * 4)[:4] + so,
                                  i=(r["k"], r["stride"], r["pad"], r["dil"], 0, 0, nx),
                                  f=(tuple(fl) + (0.0,) * 4)[:4] + tuple(cf)))
                else:
                    fwd.append(mk(KXK, ins=(x,), out=out, p=(pw, cb) + pbn, ws=(tuple(wsl) + (-1, -1, -1))[:3] + (self._abs(r["wt"]),),
                                  i=(r["stride"],), f=fl))
            elif k == EW:
                if r.get("lazy") or r.get("fwd_fused"):
                    continue
                if "flat" in r:
                    # a lazy sum that had to be materialised after all (a reader that cannot add on load): launched on the
                    # flattened operand list -- an operand that is itself a lazy sum has never been written
                    fl = r["flat"]
                    fwd.append(mk(EW, ins=[t for t, _ in fl], out=r["out"], i=(len(fl), 1), f=(r["slope"], 0.0, 0.0, 0.0) + tuple(c for _, c in fl)))
                elif r.get("mode") or r.get("coefs") is not None:
                    cf = r["coefs"] or [1.0] * len(r["srcs"])
                    fwd.append(mk(EW, ins=r["srcs"], out=r["out"], i=(len(r["srcs"]), 1, r.get("mode", 0)),
                                  f=(r["slope"], 0.0, 0.0, 0.0) + tuple(cf)))
                else:
                    fwd.append(mk(EW, ins=r["srcs"], out=r["out"], i=(len(r["srcs"]),), f=(r["slope"],)))
            elif k == SHUFFLE:
                fwd.append(mk(SHUFFLE, ins=(r["a"], r["b"]), out=r["out"]))
            elif k == MAXPOOL:
                fwd.append(mk(MAXPOOL, ins=(r["x"],), out=r["out"]))
            elif k == AVGPOOL:
                ob = self.bufs[r["out"].buf]
                fwd.append(mk(AVGPOOL, ins=(r["x"],), ws=(ob.off["data"],), i=(r["OH"], r["OW"], 0, ob.C, r["out"].coff)))
            elif k == TABLE_FILL:
                fwd.append(mk(TABLE_FILL, out=r["out"], p=(self._p(r["bias"]),), i=(1,), f=(1.0, 0.0, r["slope"])))
            elif k == CA_MLP and hasattr(r["ca"], "rbr_reparam"):
                y, ca = r["y"], r["ca"]
                if self.with_backward:
                    raise _lib.LhnError("deployed ChannelAttension is inference-only")
                fwd.append(mk(AVGPOOL, ins=(y,), ws=(self._abs(r["pooled"]),), i=(3, 3, 1)))
                fwd.append(mk(CA_MLP, out=y,
                              p=(self._p(ca.rbr_reparam.weight), -1, self._p(ca.rbr_reparam.bias), -1, -1, -1,
                                 self._p(ca.conv1x1[1].weight), self._p(ca.conv1x1[1].bias),
                                 self._p(ca.conv1x1[3].weight), self._p(ca.conv1x1[3].bias)),
                              ws=(self._abs(r["pooled"]), self._abs(r["save"]), self._abs(r["mask"])),
                              f=(1e-5, 0.1)))
            elif k == SE_MLP:
                y, dn, up = r["y"], r["down"], r["up"]
                fwd.append(mk(AVGPOOL, ins=(y,), ws=(self._abs(r["pooled"]),), i=(1, 1, 1)))
                fwd.append(mk(SE_MLP, out=y, p=(self._p(dn.weight), self._p(dn.bias), self._p(up.weight), self._p(up.bias)),
                              ws=(self._abs(r["pooled"]), self._abs(r["save"])), i=(r["J"], r["mode"])))
            elif k == ATT_MLP:
                y, att = r["y"], r["att"]
                bn, dw, lin = att[1], att[3], att[6]
                fwd.append(mk(AVGPOOL, ins=(y,), ws=(self._abs(r["pooled"]),), i=(3, 3, 1)))
                fwd.append(mk(ATT_MLP, out=y,
                              p=(self._p(bn.weight), self._p(bn.bias), self._p(bn.running_mean), self._p(bn.running_var),
                                 self._p(bn.num_batches_tracked), self._p(dw.weight), self._p(dw.bias),
                                 self._p(lin.weight), self._p(lin.bias)),
                              ws=(self._abs(r["pooled"]), self._abs(r["save"]), self._abs(r["mask"]), self._abs(r["gsum"])),
                              f=(bn.eps, bn.momentum)))
            elif k == CA_MLP:
                y, ca = r["y"], r["ca"]
                sl = r.get("bnslices")
                pins = (y, r["copy"]["srcs"][0]) if r.get("copy") else (y,)       # second input: pass-through half copied by this launch
                if sl:      # pooling pass that also leaves M0, M1 per (n, bin, c) for the backward (lhn_avgpool_fwd4)
                    fwd.append(mk(AVGPOOL, ins=pins, ws=(self._abs(r["pooled"]), self._abs(r["pstat"])) + tuple(self._abs(q["save"]) for q in sl),
                                  i=(3, 3, 1, 0, 0) + tuple((q["out"].coff << 16) | q["out"].C for q in sl)))
                else:
                    fwd.append(mk(AVGPOOL, ins=pins, ws=(self._abs(r["pooled"]),), i=(3, 3, 1)))
                bn = ca.conv3x3.bn
                fwd.append(mk(CA_MLP, out=y,
                              p=(self._p(ca.conv3x3.conv.weight), self._p(bn.weight), self._p(bn.bias),
                                 self._p(bn.running_mean), self._p(bn.running_var), self._p(bn.num_batches_tracked),
                                 self._p(ca.conv1x1[1].weight), self._p(ca.conv1x1[1].bias),
                                 self._p(ca.conv1x1[3].weight), self._p(ca.conv1x1[3].bias)),
                              ws=(self._abs(r["pooled"]), self._abs(r["save"]), self._abs(r["mask"]), self._abs(r["gsum"])),
                              f=(bn.eps, bn.momentum)))
            else:
                raise AssertionError(k)
        # (round 2's deferred finalize -- the first reader of a convolution folding the replicated statistics in its prologue --
        # was measured slower than the separate ~5 us launch and is gone: DESIGN.md section 5.2; the kernels' lhn_pend inputs stay
        # empty)
        # ---------------- backward
        if self.with_backward:
            written = {}
            self._needs_zero_grad = set()
            body = []
            # Gradient aliasing: a source of a plain (slope 1, same-size, whole-buffer) combine whose ONLY reader is
            # that combine receives exactly d(out) -- its gradient buffer becomes an alias of the output's and the
            # copy launch disappears (MSRB: the gated branch buffer of every residual add).
            uses = {}
            for r in self.recs:
                for t in ([r["x"]] if r["op"] in (PW, DW, KXK, MAXPOOL, AVGPOOL) else r["srcs"] if r["op"] == EW else
                          [r["a"], r["b"]] if r["op"] == SHUFFLE else []):
                    uses.setdefault(t.buf, []).append(r)
            aliased = set()
            self._alias_of = {}
            for r in reversed(self.recs):
                if r["op"] != EW or r["slope"] != 1.0 or r.get("mode") or r.get("coefs") is not None:
                    continue
                out = r["out"]
                ob = self.bufs[out.buf]
                if out.coff != 0 or out.C != ob.C or ob.gate or ob.dpool:
                    continue
                for t in r["srcs"]:
                    if t.buf < 0 or t.buf == out.buf or t.buf in aliased or len(uses.get(t.buf, ())) != 1:
                        continue
                    tb = self.bufs[t.buf]
                    if (self.in_ref is not None and t.buf == self.in_ref.buf) or t.coff != 0 or t.C != tb.C:
                        continue
                    if (t.H, t.W) != (out.H, out.W) or sum(1 for q in r["srcs"] if q.buf == t.buf) != 1:
                        continue
                    tb.off["grad"] = ob.off["grad"]
                    aliased.add(t.buf)
                    self._alias_of[t.buf] = out.buf
            self.grad_aliases = len(aliased)
            materialised = set()
            # Gradient addends: a plain residual sum O = S + ... hands d(O) to every source.  When S's other readers are
            # tiled depthwise convolutions whose input slices tile S exactly (MSRB: `out` feeds the two dilated 3x3 halves
            # and the running sum, litehourglass.py:41-49), those convolutions' backward kernels add d(O) while storing
            # their dx (include/lhn.h: lhn_conv_dw_bwd3) and the sum's copy / accumulate pass over S's gradient disappears.
            pending_add, self.grad_addends = {}, 0
            if os.environ.get("LHN_GRAD_ADDENDS", "1") != "0":
                order = {id(q): j for j, q in enumerate(self.recs)}

                def plain_sum(q):
                    return q["op"] == EW and q["slope"] == 1.0 and not q.get("mode") and q.get("coefs") is None and \
                        not isinstance(q["out"], TCat)

                for sb, rd in uses.items():
                    if sb < 0 or sb in aliased or sb == self._no_grad_buf:
                        continue
                    sums = [q for q in rd if plain_sum(q)]
                    dws = [q for q in rd if not plain_sum(q)]
                    if not sums or not dws or len(sums) > 2:
                        continue
                    if not all(q["op"] == DW and q["conv"].weight is not None and q["stride"] == 1 and q["k"] == 3 and
                               q["pad"] == q["dil"] and q["x"].C % 32 == 0 and q["x"].W >= 8 and not isinstance(q["x"], TCat)
                               for q in dws):
                        continue
                    if min(order[id(q)] for q in sums) < max(order[id(q)] for q in dws):
                        continue
                    spans = sorted((q["x"].coff, q["x"].coff + q["x"].C) for q in dws)
                    lo, hi = spans[0][0], spans[-1][1]
                    if any(a[1] != b[0] for a, b in zip(spans, spans[1:])):
                        continue
                    adds = []
                    for q in sums:
                        mine = [t for t in q["srcs"] if not isinstance(t, TCat) and t.buf == sb]
                        o = q["out"]
                        if len(mine) != 1 or (mine[0].coff, mine[0].C) != (lo, hi - lo) or (mine[0].H, mine[0].W) != (o.H, o.W) or \
                                self.bufs[o.buf].C != self.bufs[sb].C:
                            break
                        adds.append((id(q), o.buf, o.coff - lo))
                    else:
                        pending_add[sb] = adds
            fuse_sums = os.environ.get("LHN_FUSE_BN_SUMS", "1") != "0"
            self.fused_bn_sums = 0
            for r in self.recs:
                r.pop("sums_by_reader", None)
                r.pop("sums_by_readers", None)
            # Reader-side BatchNorm sums, general form (include/lhn.h: lhn_bnsum): du is linear in dz and dz is the sum of what the
            # readers' backward kernels hand back, so when EVERY reader of a convolution + BatchNorm output can add its part
            # (elementwise combines, pools, the fused 1x1 backward) the producer's lhn_bn_bwd_reduce pass is not launched.
            # bns[(id(reader record), buf, coff, C)] = (producer record, channel offset inside the producer's BatchNorm)
            bns = {}
            self.reader_bn_sums = 0
            if os.environ.get("LHN_READER_BN_SUMS", "1") != "0":
                for P in self.recs:
                    if P["op"] not in (STEM, PW, DW, KXK) or P["bn"] is None or P.get("sums_by_ca") or isinstance(P["out"], TCat):
                        continue
                    o = P["out"]
                    if o.buf < 0 or P.get("wrc", (0, 0))[0] or P.get("bn_repeat", 1) != 1 or o.buf in aliased:
                        continue
                    ob = self.bufs[o.buf]
                    if ob.gate or ob.dpool or ob.lazy is not None or (self.out_ref is not None and o.buf == self.out_ref.buf):
                        continue
                    lo, hi = o.coff, o.coff + o.C
                    mine, ok = [], True
                    for q in uses.get(o.buf, ()):
                        views = [q["x"]] if q["op"] in (PW, DW, KXK, MAXPOOL, AVGPOOL) else q["srcs"] if q["op"] == EW else [q["a"], q["b"]]
                        for t in views:
                            if t.buf != o.buf or t.coff + t.C <= lo or hi <= t.coff:
                                continue
                            if not (lo <= t.coff and t.coff + t.C <= hi):
                                ok = False
                            elif q["op"] == EW:
                                ok = ok and not q.get("lazy") and "flat" not in q and not q.get("mode") and q.get("coefs") is None and \
                                    q["slope"] not in (SLOPE_SILU, SLOPE_RELU_SIGMOID) and not isinstance(q["out"], TCat) and \
                                    not any(a[0] == id(q) for a in pending_add.get(t.buf, ())) and \
                                    q["out"].H % t.H == 0 and q["out"].W % t.W == 0
                            elif q["op"] in (MAXPOOL, AVGPOOL):
                                pass
                            elif q["op"] == PW:
                                # (the fused 1x1 backward kernel keeps the shapes whose TILES stay below 64 x 128: csrc/k_conv_pw.hip)
                                ci_t = 32 if t.C <= 32 else 64 if t.C <= 64 else 128
                                nto = (q["out"].C + 31) // 32
                                nto = 4 if nto == 3 else nto
                                ok = ok and q["stride"] == 1 and not q.get("nchw") and q.get("xs") is None and t.C <= 128 and \
                                    q["out"].C <= 128 and ci_t * nto * 32 < 64 * 128 and not q["wrc"][1] and not q["wrc"][0] and \
                                    (t.coff, t.C) == (q["x"].coff, q["x"].C)
                            else:
                                ok = False
                            mine.append((q, t))
                    if ok and mine:
                        P["sums_by_readers"] = True
                        self.reader_bn_sums += 1
                        for q, t in mine:
                            bns[(id(q), t.buf, t.coff, t.C)] = (P, t.coff - lo)

            def bns_of(q, t):
                """(ws pair, (C, coff)) of the BatchNorm sums reader record q adds for its input view t, or ((-1, -1), (0, 0))."""
                e = bns.get((id(q), t.buf, t.coff, t.C))
                if e is None:
                    return (-1, -1), (0, 0)
                P, off = e
                return (self._abs(P["sums"]), self._abs(P["save"])), (P["out"].C, off)
            # ---- readers of one tensor whose gradients meet in ONE store (lhn_grad_adds): a 2x2 max-pool, an adaptive average pool
            # and a plain same-resolution sum reading the same view (the skip tensor of an hourglass level, litehourglass.py:139-163)
            # -- the max-pool's backward, which runs last, takes the other two gradients on the way (LHN_POOL_GRAD_ADDS=0: three
            # read-modify-write passes over d(x) as before)
            fused_mp, skip_ew_src, skip_ap = {}, set(), set()
            if os.environ.get("LHN_POOL_GRAD_ADDS", "1") != "0":
                order = {id(q): i for i, q in enumerate(self.recs)}
                for mp in self.recs:
                    if mp["op"] != MAXPOOL or isinstance(mp["x"], TCat) or mp["x"].H % 2 or mp["x"].W % 2 or mp["x"].buf == self._no_grad_buf:
                        continue
                    X = mp["x"]

                    def same(t, X=X):
                        return not isinstance(t, TCat) and t.buf == X.buf and t.coff == X.coff and t.C == X.C
                    ap = next((q for q in self.recs if q["op"] == AVGPOOL and "OH" in q and same(q["x"]) and order[id(q)] > order[id(mp)]
                               and id(q) not in skip_ap), None)
                    ew = None
                    for q in self.recs:
                        if q["op"] != EW or q.get("lazy") or q.get("fwd_fused") or "flat" in q or q.get("mode") or q.get("coefs") is not None:
                            continue
                        if order[id(q)] < order[id(mp)] or float(q["slope"]) != 1.0 or isinstance(q["out"], TCat):
                            continue
                        ob = self.bufs[q["out"].buf]
                        idx = [j for j, t in enumerate(q["srcs"]) if same(t)]
                        if ob.gate or ob.dpool or len(idx) != 1 or (q["srcs"][idx[0]].H, q["srcs"][idx[0]].W) != (q["out"].H, q["out"].W):
                            continue
                        if (id(q), idx[0]) in skip_ew_src or any(a[0] == id(q) for a in pending_add.get(X.buf, ())):
                            continue
                        ew = (q, idx[0])
                        break
                    if ap is None and ew is None:
                        continue
                    keys = [bns_of(mp, X)] + ([bns_of(ap, X)] if ap else []) + ([bns_of(ew[0], X)] if ew else [])
                    if any(kk != keys[0] for kk in keys):      # the producer's BatchNorm sums: all of x's readers or none
                        continue
                    fused_mp[id(mp)] = (ew, ap)
                    if ew:
                        skip_ew_src.add((id(ew[0]), ew[1]))
                    if ap:
                        skip_ap.add(id(ap))
            self.pool_grad_adds = len(fused_mp)
            for r in reversed(self.recs):
                k = r["op"]
                if k in (STEM, PW, DW, KXK, EW, SHUFFLE, MAXPOOL, AVGPOOL) and not (k == PW and r.get("nchw")) and \
                        not (self.out_ref is not None and not isinstance(r["out"], TCat) and r["out"].buf == self.out_ref.buf):
                    self._covered(written, r["out"])      # (the block output's gradient is written by the engine, not by an op)
                if k in (STEM, PW, DW, KXK):
                    conv, bn, x, out = r["conv"], r["bn"], r["x"], r["out"]
                    pw = self._p(conv.weight)
                    use_coef = 1 if bn is not None else 0
                    lz = self.bufs[x.buf].lazy if x.buf >= 0 else None
                    if lz is not None and x.buf not in materialised and x.buf not in self._sum_written:
                        # the weight gradient needs the summed input: written once per backward, whole buffer
                        materialised.add(x.buf)
                        fl = lz["flat"]
                        body.append(mk(EW, ins=[t for t, _ in fl], out=lz["out"], i=(len(fl), 1),
                                       f=(1.0, 0.0, 0.0, 0.0) + tuple(c for _, c in fl)))
                    if bn is not None:
                        body.append(mk(BN_BWD, out=out, p=(self._p(bn.weight), self._p(bn.weight), self._p(bn.bias)),
                                       ws=(self._abs(r["sums"]), self._abs(r["save"]), self._abs(r["bcnt"])),
                                       i=(r["wrc"][0] if k == PW else 0, 1 if (r.get("sums_by_reader") or r.get("sums_by_ca") or r.get("sums_by_readers")) else 0)))
                    if k == STEM:
                        body.append(mk(STEM_BWD, out=out, p=(pw, pw), i=(r["k"], r["stride"], r["pad"], x.H, x.W, use_coef)))
                        continue
                    need_dx = x.buf != self._no_grad_buf
                    mode = self._grad_mode(written, x) if need_dx else 0
                    if k == PW:
                        if r["stride"] != 1 and mode == 1:   # strided dgrad touches a subset of pixels
                            self._needs_zero_grad.add(x.buf)
                            mode = 2
                        o = TRef(-1, 0, out.C, out.H, out.W) if r["nchw"] else out
                        # a bias in front of a train-mode BatchNorm has an identically zero gradient
                        bw, bc = bns_of(r, x) if need_dx else ((-1, -1), (0, 0))
                        body.append(mk(PW_BWD, ins=(x,), out=o, p=(pw, pw, self._p(conv.bias) if bn is None else -1), ws=bw,
                                       i=(r["stride"], 1 if r["nchw"] else 0, mode, r["wrc"][0], r["wrc"][1], use_coef,
                                          r["stack"][0], r["stack"][1]), f=(0.0,) * 6 + (float(bc[0]), float(bc[1]))))
                    elif k == DW:
                        # the producer's BatchNorm-backward sums ride in this kernel when it is the only reader of x
                        # (RepBasicUnit 1x1 -> 3x3 depthwise; include/lhn.h: lhn_conv_dw_bwd2)
                        prod, xb = None, self.bufs[x.buf]
                        if (fuse_sums and mode == 1 and r["k"] == 3 and r["stride"] == 1 and r["pad"] == 1 and r["dil"] == 1 and
                                x.C % 32 == 0 and x.W >= 8 and not xb.gate and not xb.dpool and xb.lazy is None and
                                conv.weight is not None and len(uses.get(x.buf, ())) == 1 and x.buf not in aliased):
                            for q in self.recs:
                                if q["op"] in (STEM, PW, DW, KXK) and q["bn"] is not None and q["out"].buf == x.buf and \
                                        (q["out"].coff, q["out"].C) == (x.coff, x.C) and not q.get("wrc", (0, 0))[0] and q.get("bn_repeat", 1) == 1:
                                    prod = q
                        adds = pending_add.get(x.buf)
                        if adds and need_dx:
                            self.grad_addends += 1
                            offs = [self.bufs[ob].off["grad"] + 4 * sh for _, ob, sh in adds]
                            body.append(mk(DW_BWD, ins=(x,), out=out, p=(pw, pw), ws=(-1, -1, offs[0], offs[1] if len(offs) > 1 else -1),
                                           i=(r["k"], r["stride"], r["pad"], r["dil"], mode, use_coef)))
                        elif prod is not None:
                            prod["sums_by_reader"] = True
                            self.fused_bn_sums += 1
                            body.append(mk(DW_BWD, ins=(x,), out=out, p=(pw, pw), ws=(-1, -1, -1, -1, self._abs(prod["sums"]), self._abs(prod["save"])),
                                           i=(r["k"], r["stride"], r["pad"], r["dil"], mode, use_coef, x.C, 0)))
                        else:
                            body.append(mk(DW_BWD, ins=(x,), out=out, p=(pw, pw),
                                           i=(r["k"], r["stride"], r["pad"], r["dil"], mode, use_coef)))
                    else:
                        body.append(mk(KXK_BWD, ins=(x,), out=out, p=(pw, pw), ws=(-1, -1, -1, self._abs(r["wt"])),
                                       i=(r["stride"], 0, mode, 0, 0, use_coef)))
                elif k == EW and (r.get("mode") or r.get("coefs") is not None):
                    md, srcs = r.get("mode", 0), r["srcs"]
                    if r.get("coefs") is not None and any(c != 1.0 for c in r["coefs"]):
                        raise _lib.LhnError("a combine with coefficients is forward-only")
                    for j, s in enumerate(srcs):
                        if s.buf == self._no_grad_buf:
                            continue
                        gm = self._grad_mode(written, s)
                        acc = 1 if gm == 2 else 0
                        if md & EW_MUL:
                            body.append(mk(EW_BWD, ins=(s, srcs[1 - j]), out=r["out"], i=(acc, 1), f=(r["slope"],)))
                        elif (md & EW_BILINEAR) and (s.H, s.W) != (r["out"].H, r["out"].W):
                            body.append(mk(EW_BWD, ins=(s,), out=r["out"], i=(acc, 2), f=(r["slope"],)))
                        else:
                            body.append(mk(EW_BWD, ins=(s,), out=r["out"], i=(acc,), f=(r["slope"],)))
                elif k == EW:
                    todo = []
                    for j, s in enumerate(r["srcs"]):
                        if s.buf == self._no_grad_buf or (s.buf in aliased and self.bufs[s.buf].off["grad"] == self.bufs[r["out"].buf].off["grad"]):
                            continue
                        if (id(r), j) in skip_ew_src:
                            continue                # d(out) joins s's gradient inside the max-pool backward that reads s (fused_mp)
                        if any(a[0] == id(r) for a in pending_add.get(s.buf, ())):
                            continue                # d(out) joins s's gradient inside the depthwise backward kernels that read s
                        mode = self._grad_mode(written, s)
                        bw, bc = bns_of(r, s)
                        todo.append((s, 1 if mode == 2 else 0, bw, bc))
                    # sources of the destination's own resolution (residual sums) share ONE pass over d(out) / out (lhn_ew_bwd_multi)
                    o_ = r["out"]
                    same = [t for t in todo if not isinstance(t[0], TCat) and (t[0].H, t[0].W, t[0].C) == (o_.H, o_.W, o_.C)]
                    if os.environ.get("LHN_EW_BWD_MULTI", "1") == "0" or len(same) < 2 or r["slope"] in (SLOPE_SILU, SLOPE_RELU_SIGMOID):
                        same = []
                    groups, rest_same = [], list(same)
                    while len(rest_same) >= 2:
                        take = 3 if len(rest_same) != 4 else 2
                        groups.append(rest_same[:take])
                        rest_same = rest_same[take:]
                    grouped = {id(t[0]) for g in groups for t in g}
                    for g in groups:
                        ws_, iv, fv = [], [g[0][1], 3, g[1][1], g[2][1] if len(g) > 2 else 0], [r["slope"], 0.0, 0.0, 0.0, 0.0, 0.0]
                        cc = []
                        for q in range(3):
                            if q < len(g):
                                ws_ += list(g[q][2])
                                cc.append(g[q][3])
                            else:
                                ws_ += [-1, -1]
                                cc.append((0, 0))
                        iv += [cc[0][0], cc[0][1], cc[1][0], cc[1][1]]
                        fv[4], fv[5] = float(cc[2][0]), float(cc[2][1])
                        body.append(mk(EW_BWD, ins=tuple(t[0] for t in g), out=o_, ws=tuple(ws_), i=tuple(iv), f=tuple(fv)))
                        self.ew_bwd_multi = getattr(self, "ew_bwd_multi", 0) + 1
                    for s, acc, bw, bc in todo:
                        if id(s) in grouped:
                            continue
                        body.append(mk(EW_BWD, ins=(s,), out=o_, ws=bw, i=(acc, 0, 0, 0, bc[0], bc[1]), f=(r["slope"],)))
                elif k == SHUFFLE:
                    ma = 0 if r["a"].buf == self._no_grad_buf else self._grad_mode(written, r["a"])
                    mb = 0 if r["b"].buf == self._no_grad_buf else self._grad_mode(written, r["b"])
                    body.append(mk(SHUFFLE_BWD, ins=(r["a"], r["b"]), out=r["out"], i=(ma, mb)))
                elif k == MAXPOOL:
                    mode = self._grad_mode(written, r["x"])
                    bw, bc = bns_of(r, r["x"])
                    ew, ap = fused_mp.get(id(r), (None, None))
                    sw, si, pw_, pi = -1, (0, 0), -1, (0, 0, 0)
                    if ew is not None:
                        eo = ew[0]["out"]
                        sw, si = self.bufs[eo.buf].off["grad"], (self.bufs[eo.buf].C, eo.coff)
                    if ap is not None:
                        ob = self.bufs[ap["out"].buf]
                        pw_, pi = ob.off["grad"], (ob.C, ap["out"].coff, (ap["OH"] << 16) | ap["OW"])
                    body.append(mk(MAXPOOL_BWD, ins=(r["x"],), out=r["out"], ws=bw + (sw, pw_),
                                   i=(1 if mode == 2 else 0, si[0], si[1], pi[0], bc[0], bc[1], pi[1], pi[2])))
                elif k == AVGPOOL:
                    if id(r) in skip_ap:
                        continue                    # its gradient joins d(x) inside the max-pool backward (fused_mp)
                    mode = self._grad_mode(written, r["x"])
                    ob = self.bufs[r["out"].buf]
                    bw, bc = bns_of(r, r["x"])
                    body.append(mk(AVGPOOL_BWD, ins=(r["x"],), ws=(ob.off["grad"],) + bw,
                                   i=(r["OH"], r["OW"], 1 if mode == 2 else 0, ob.C, r["out"].coff, bc[0], bc[1])))
                elif k == SE_MLP:
                    y, dn, up = r["y"], r["down"], r["up"]
                    body.append(mk(GATE_REDUCE, out=y, ws=(-1, -1, -1, self._abs(r["dgate"]))))
                    body.append(mk(SE_MLP_BWD, out=y,
                                   p=(self._p(dn.weight), self._p(up.weight), self._p(dn.weight),
                                      self._p(dn.bias), self._p(up.weight), self._p(up.bias)),
                                   ws=(self._abs(r["pooled"]), self._abs(r["save"]), -1, self._abs(r["dgate"])), i=(r["J"], r["mode"])))
                elif k == ATT_MLP:
                    y, att = r["y"], r["att"]
                    bn, dw, lin = att[1], att[3], att[6]
                    body.append(mk(GATE_REDUCE, out=y, ws=(-1, -1, -1, self._abs(r["dgate"]))))
                    body.append(mk(ATT_MLP_BWD, out=y,
                                   p=(self._p(bn.weight), self._p(bn.bias), self._p(dw.weight), self._p(lin.weight),
                                      self._p(bn.weight), self._p(bn.bias), self._p(dw.weight), self._p(dw.bias),
                                      self._p(lin.weight), self._p(lin.bias)),
                                   ws=(self._abs(r["pooled"]), self._abs(r["save"]), self._abs(r["mask"]),
                                       self._abs(r["dgate"]), self._abs(r["gsum_b"]))))
                elif k == CA_MLP:
                    y, ca = r["y"], r["ca"]
                    bn = ca.conv3x3.bn
                    sl = r.get("bnslices") or []
                    pk = tuple((q["out"].coff << 16) | q["out"].C for q in sl)
                    sv = (tuple(self._abs(q["save"]) for q in sl) + (-1, -1))[:2]
                    sm = (tuple(self._abs(q["sums"]) for q in sl) + (-1, -1))[:2]
                    body.append(mk(GATE_REDUCE, out=y, ws=(-1, -1, -1, self._abs(r["dgate"]), 1 if sl else -1) + sv, i=pk))
                    body.append(mk(CA_MLP_BWD, out=y,
                                   p=(self._p(ca.conv3x3.conv.weight), self._p(bn.weight), self._p(ca.conv1x1[1].weight),
                                      self._p(ca.conv1x1[3].weight),
                                      self._p(ca.conv3x3.conv.weight), self._p(bn.weight), self._p(bn.bias),
                                      self._p(ca.conv1x1[1].weight), self._p(ca.conv1x1[1].bias),
                                      self._p(ca.conv1x1[3].weight), self._p(ca.conv1x1[3].bias)),
                                   ws=(self._abs(r["pooled"]), self._abs(r["save"]), self._abs(r["mask"]),
                                       self._abs(r["dgate"]), self._abs(r["gsum_b"]), self._abs(r.get("pstat"))) + sv + sm, i=pk))
            if self.ar["zb"].size:
                bwd.append(mk(MEMSET, ws=(self.arena_base["zb"], self.ar["zb"].size)))
            for b in sorted(self._needs_zero_grad):
                rec = self.bufs[b]
                bwd.append(mk(MEMSET, ws=(rec.off["grad"], N * rec.H * rec.W * rec.C * 4)))
            if self._needs_zero_grad:
                # every write into a zero-initialised gradient buffer must accumulate
                for o in body:
                    if o.kind in (PW_BWD, KXK_BWD) and o.in_buf[0] in self._needs_zero_grad and o.i[2]:
                        o.i[2] = 2
                    elif o.kind == DW_BWD and o.in_buf[0] in self._needs_zero_grad and o.i[4]:
                        o.i[4] = 2
                    elif o.kind in (EW_BWD, MAXPOOL_BWD) and o.in_buf[0] in self._needs_zero_grad:
                        o.i[0] = 1
                    elif o.kind == SHUFFLE_BWD:
                        for q in range(2):
                            if o.in_buf[q] in self._needs_zero_grad and o.i[q]:
                                o.i[q] = 2
                    elif o.kind == AVGPOOL_BWD and o.in_buf[0] in self._needs_zero_grad:
                        o.i[2] = 1
            bwd += body
        # ---------------- SyncBatchNorm: (op index, byte offset, number of doubles, replicated layout?) of every statistics buffer that has
        # to be all-reduced between half-step 2*i and 2*i+1 of lhn_plan_run_range
        self.sync_points = {0: [], 1: []}
        for phase, lst in ((0, fwd), (1, bwd)):
            for i, o in enumerate(lst):
                if o.kind in (STEM, PW, DW, KXK) and o.p[2] >= 0 and o.ws[0] >= 0:
                    self.sync_points[0].append((i, o.ws[0], STAT_REPLICAS * 2 * o.out_C, True))
                elif o.kind in (CA_MLP, ATT_MLP) and o.ws[3] >= 0:
                    self.sync_points[0].append((i, o.ws[3], 2 * o.out_C, False))
                elif o.kind == BN_BWD:
                    self.sync_points[1].append((i, o.ws[0], STAT_REPLICAS * 2 * o.out_C, True))
                elif o.kind in (CA_MLP_BWD, ATT_MLP_BWD) and o.ws[4] >= 0:
                    self.sync_points[1].append((i, o.ws[4], 2 * o.out_C, False))
        # ---------------- C arrays
        cb = (Buf * len(self.bufs))()
        for j, b in enumerate(self.bufs):
            cb[j].data_off = b.off["data"]
            cb[j].table_off = b.off["table"]
            cb[j].gate_off = b.off.get("gate", -1)
            cb[j].grad_off = b.off.get("grad", -1)
            cb[j].dpool_off = b.off.get("dpool", -1)
            cb[j].coef_off = b.off.get("coef", -1)
            cb[j].N, cb[j].H, cb[j].W, cb[j].C = N, b.H, b.W, b.C
        cf = (Op * len(fwd))(*fwd)
        cbw = (Op * max(1, len(bwd)))(*bwd) if bwd else None
        return cb, cf, cbw, len(fwd), len(bwd)

    _no_grad_buf = -1   # the image never needs a gradient


_SIDE_STREAMS = {}


def _graph_default(n_ops):
    """hipGraph replay of a plan's launch sequence (LHN_RUN_GRAPH): only with LHN_GRAPH=1.  Measured on MI355X (round 3, same
    box, Lite-HRNet-18 at batch 64, 663 forward ops / 2,600 launches per step): replay 38.83 ms per step and 13.22 ms per
    forward against 37.16 / 12.43 ms with plain launches -- outside the profiler the host keeps up with the queue, and a
    graph launch adds its fixed cost per phase.  So replay is NOT a default for any plan size."""
    return os.environ.get("LHN_GRAPH", "") == "1"


def _side_stream(device):
    """One non-default stream per device: the legacy default stream cannot be captured into a hipGraph."""
    key = torch.device(device).index
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


_TRAIN_RUNS = 0      # bumped by every train-mode forward of any plan in this process (see CompiledPlan.run)
_TABLE_EPOCH = 0     # bumped by invalidate_tables(): state changed where neither torch's version counters nor data pointers see it


def invalidate_tables():
    """Every plan's cached eval-mode BatchNorm tables / deployed biases become stale.  Call after changing parameters or
    running statistics through `.data` (p.data.mul_(..), EMA / weight surgery): such writes bump no version counter.
    litehandnet_amd.train.Trainer.step and FlatParams.broadcast call it themselves."""
    global _TABLE_EPOCH
    _TABLE_EPOCH += 1


class CompiledPlan:
    """Owns the C plan + the workspace arena (a torch uint8 tensor) for one (module, input shape)."""

    def __init__(self, pb: PlanBuilder, state_tensors, device):
        self.pb = pb
        cb, cf, cbw, nf, nb = pb.finalize()
        self._keep = (cb, cf, cbw)
        L = _lib.lib()
        self.handle = L.lhn_plan_create(cb, len(pb.bufs), cf, nf, cbw, nb)
        if not self.handle:
            raise _lib.LhnError("lhn_plan_create: " + L.lhn_last_error().decode())
        self.n_fwd, self.n_bwd = nf, nb
        self.use_graph = _graph_default(nf)
        self.ws = torch.empty(pb.total_bytes, dtype=torch.uint8, device=device)
        # tables start as the identity transform (scale 1, shift 0, slope 1); FINALIZE overwrites BN slices
        for b in pb.bufs:
            t = self.view_f32(b.off["table"], 3 * b.C).view(3, b.C)
            t[0].fill_(1.0)
            t[1].zero_()
            t[2].fill_(1.0)
            if "coef" in b.off:     # dy = A du + B y + C: identity until a BatchNorm backward writes its slice (padded
                c = self.view_f32(b.off["coef"], 3 * b.C).view(3, b.C)      # channels of a 7 / 17 / 37-wide one never are)
                c[0].fill_(1.0)
                c[1].zero_()
                c[2].zero_()
        self.state_tensors = state_tensors
        self._params = (C.c_void_p * len(state_tensors))()
        self._grads = (C.c_void_p * len(state_tensors))()
        self._io = (C.c_void_p * 2)()
        self._table_sig = None
        self.fwd_serial = 0            # forwards run on this workspace so far
        self.bwd_serial = -1           # serial of the forward whose backward has already consumed the workspace
        self.mask_view = None
        # dropout masks (Dropout2d of ChannelAttension, common.py:57; Dropout of mynet's attention, pose_hg_ms_att.py:171):
        # one [N, C] slice per attention module, values 0 or 1/keep.  `mask_fn(plan)`, when set, fills them instead of the
        # default bernoulli_ draw -- parity tests feed the oracle the same masks.
        self.mask_fn = None            # plan-level override; otherwise the owning engine's `mask_fn` as it is at run time
        self.engine = None
        self.mask_slices = []
        if pb.ar["mask"].size:
            self.mask_view = self.view_f32(pb.arena_base["mask"], pb.ar["mask"].size // 4)
            for r in pb.recs:
                if r.get("mask") is not None:
                    mod = r.get("ca", r.get("att"))
                    Cc = r["y"].C
                    self.mask_slices.append((mod, self.view_f32(pb._abs(r["mask"]), pb.N * Cc).view(pb.N, Cc)))

    def view_f32(self, byte_off, numel):
        return self.ws[byte_off:byte_off + numel * 4].view(torch.float32)

    def buf_data(self, ref, grad=False):
        b = self.pb.bufs[ref.buf]
        v = self.view_f32(b.off["grad" if grad else "data"], self.pb.N * b.H * b.W * b.C)
        return v.view(self.pb.N, b.H, b.W, b.C)[..., ref.coff:ref.coff + ref.C]

    def refresh_params(self):
        for j, t in enumerate(self.state_tensors):
            self._params[j] = t.data_ptr()

    def set_grads(self, grad_tensors):
        for j, g in enumerate(grad_tensors):
            self._grads[j] = 0 if g is None else g.data_ptr()

    def run(self, phase, io0, io1, training, grad_replicas=1, grad_rep_stride=0, sync=None):
        """sync = (world, all_reduce_fn) runs the phase in SyncBatchNorm mode: the launch sequence is cut at every
        statistics buffer, `all_reduce_fn(float64 view)` sums it over the ranks, statistics count N*world samples."""
        self._io[0] = 0 if io0 is None else io0.data_ptr()
        self._io[1] = 0 if io1 is None else io1.data_ptr()
        global _TRAIN_RUNS
        if phase == 0:
            self.fwd_serial += 1       # the workspace (activations, BatchNorm saves, masks, gates) now belongs to THIS forward
        if phase == 0 and training:
            _TRAIN_RUNS += 1           # running statistics are about to move: every plan's eval tables become stale
            self._table_sig = None
        if phase == 0 and training and self.mask_view is not None:
            fn = self.mask_fn if self.mask_fn is not None else getattr(self.engine, "mask_fn", None)
            if fn is not None:
                fn(self)
            else:
                keep = 1.0 - self.pb.p_drop
                self.mask_view.bernoulli_(keep).mul_(1.0 / keep)
        L = _lib.lib()
        if sync is not None and training and sync[0] > 1:
            world, allreduce = sync
            nsteps = 2 * (self.n_fwd if phase == 0 else self.n_bwd)

            def run_range(b, e):
                rc = L.lhn_plan_run_range(C.c_void_p(self.handle), phase, C.c_int64(b), C.c_int64(e), _lib.ptr(self.ws),
                                          self._params, self._grads, self._io, 1, int(grad_replicas),
                                          C.c_int64(int(grad_rep_stride)), C.c_double(float(world)), C.c_float(1.0 / world),
                                          _lib.stream())
                _lib.check(rc, "lhn_plan_run_range")
            begin = 0
            self.sync_wire_doubles = 0
            for oi, off, n, replicated in self.pb.sync_points[phase]:
                run_range(begin, 2 * oi + 1)
                if replicated:
                    # replicated [32][2][C] sums: folded on the device first, only [2][C] doubles cross the wire
                    m = n // STAT_REPLICAS
                    _lib.check(L.lhn_fold_stat_replicas(C.c_void_p(self.ws.data_ptr() + off), C.c_int64(m), STAT_REPLICAS, _lib.stream()),
                               "lhn_fold_stat_replicas")
                    n = m
                allreduce(self.ws[off:off + 8 * n].view(torch.float64))
                self.sync_wire_doubles += n
                begin = 2 * oi + 1
            run_range(begin, nsteps)
            return
        mode = 1 if training else 0
        if phase == 0:
            # eval-mode BatchNorm tables / deployed biases only depend on the parameters: rebuild them when something changed
            # (torch's per-tensor version counters catch load_state_dict / optimizer steps; our own train-mode kernels update
            # running statistics behind torch's back, so any training run anywhere invalidates every plan's tables)
            if not training and self._tables_current():
                mode |= 2              # LHN_RUN_TABLES_CURRENT
        if self.use_graph:
            # replayed as a hipGraph (LHN_RUN_GRAPH).  On torch's default stream -- which cannot be captured -- the phase runs
            # on a side stream ordered after everything queued so far, and the default stream waits for it: the caller sees
            # the usual stream semantics (no exec, no capture of the default stream).
            cur = torch.cuda.current_stream(self.ws.device)
            if cur.cuda_stream == 0:
                side = _side_stream(self.ws.device)
                side.wait_stream(cur)
                rc = L.lhn_plan_run(C.c_void_p(self.handle), phase, _lib.ptr(self.ws), self._params, self._grads, self._io,
                                    mode | 4, int(grad_replicas), C.c_int64(int(grad_rep_stride)), C.c_void_p(side.cuda_stream))
                cur.wait_stream(side)
                _lib.check(rc, "lhn_plan_run")
                return
            mode |= 4
        rc = L.lhn_plan_run(C.c_void_p(self.handle), phase, _lib.ptr(self.ws), self._params, self._grads, self._io,
                            mode, int(grad_replicas), C.c_int64(int(grad_rep_stride)), _lib.stream())
        _lib.check(rc, "lhn_plan_run")

    def _tables_current(self):
        """True when this eval run may reuse the tables the previous eval run of this plan built.  Remembers the state it
        saw: (process-wide count of train-mode runs, the invalidate_tables() epoch, torch's version counter and the data
        pointer of every parameter / buffer).  Tensors without a version counter (created under torch.inference_mode)
        disable the reuse.  Writes through `.data` are invisible to all of these: see invalidate_tables()."""
        try:
            sig = (_TRAIN_RUNS, _TABLE_EPOCH, tuple(t._version for t in self.state_tensors),
                   tuple(t.data_ptr() for t in self.state_tensors))
        except RuntimeError:
            self._table_sig = None
            return False
        same = sig == self._table_sig
        self._table_sig = sig
        return same

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().lhn_plan_destroy(C.c_void_p(self.handle))
                self.handle = None
        except Exception:
            pass

"""HIP-backed mirror of models/pose_estimation/liteHandNet/repblocks.py (train-time forward/backward).

Same constructor arguments and `state_dict()` keys as the reference classes; the arithmetic runs in
liblhn (csrc/k_conv_*.hip).  `switch_to_deploy()` folds the BatchNorms into one biased convolution per unit
(csrc/k_deploy.hip, bit-exact with the reference's fusion); the deployed form is inference-only."""
import torch
from torch import nn

from . import _lib
from .engine import PlanModule
from .plan import SLOPE_SILU


def _fold_branch(w, bn, out_w, out_b, k, cin_g, accumulate):
    """One lhn_fold_bn call: `w` is the branch kernel (None = identity branch)."""
    kb = 1 if w is None else w.shape[-1]
    for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var):
        _lib.require_device(t)
    _lib.check(_lib.lib().lhn_fold_bn(_lib.ptr(w), kb, _lib.ptr(bn.weight), _lib.ptr(bn.bias),
                                      _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var), _lib.C.c_float(bn.eps),
                                      _lib.ptr(out_w), _lib.ptr(out_b), out_w.shape[0], cin_g, k, int(accumulate),
                                      _lib.stream()), "lhn_fold_bn")


def fused_conv(like, branches):
    """A biased nn.Conv2d shaped like `like` whose kernel/bias is the sum of the BN-folded `branches`
    [(kernel-or-None, bn), ...] -- summed in the listed order (repblocks.py:180-187 adds 1x1 + identity first)."""
    w0 = like.weight
    cout, cin_g, k, _ = w0.shape
    conv = nn.Conv2d(like.in_channels, cout, like.kernel_size, like.stride, like.padding, like.dilation, like.groups,
                     bias=True, device=w0.device, dtype=w0.dtype)
    with torch.no_grad():
        for j, (w, bn) in enumerate(branches):
            _fold_branch(None if w is None else w.detach().contiguous(), bn, conv.weight, conv.bias, k, cin_g, j > 0)
    conv.weight.requires_grad_(False)
    conv.bias.requires_grad_(False)
    return conv


def conv_bn(in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1):
    """repblocks.py:8-20 -- a parameter container (`conv`, `bn`); never called directly."""
    seq = nn.Sequential()
    seq.add_module("conv", nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                     bias=False))
    seq.add_module("bn", nn.BatchNorm2d(out_channels))
    return seq


def act_slope(activation, inplace=False, positional=True):
    """Leaky slope of the unit's non-linearity.

    repblocks.py:29-30 builds `activation(inplace)` positionally, so for nn.LeakyReLU the slope IS
    float(inplace) (0 -> ReLU-like, 1 -> identity); RepBlock (:92-93) and the plain `activation()`
    calls in liteHandNet.py use the keyword / default form, i.e. the real 0.01."""
    if activation is None:
        return 1.0
    if isinstance(activation, str):
        activation = {"leakyrelu": nn.LeakyReLU, "relu": nn.ReLU, "silu": nn.SiLU}[activation.lower()]
    if activation is nn.LeakyReLU:
        return float(inplace) if positional else 0.01
    if activation is nn.ReLU:
        return 0.0
    if activation is nn.Identity:
        return 1.0
    if activation is nn.SiLU:
        return SLOPE_SILU          # not a leaky slope: applied by the elementwise combine (see RepConv.emit)
    raise _lib.LhnError(f"activation {activation} is neither a leaky slope nor SiLU")


class RepConv(PlanModule):
    """act(BN(conv(x)))  -- repblocks.py:23-44."""

    def __init__(self, in_channels, out_channels, kernel=1, stride=1, padding=0, dilation=1, groups=1, deploy=False,
                 activation=nn.LeakyReLU, inplace=False):
        super().__init__()
        self.deploy = deploy
        self.slope = act_slope(activation, inplace, positional=True)
        if deploy:
            self.rep_conv = nn.Conv2d(in_channels, out_channels, kernel, stride, padding, dilation, groups, bias=True)
        else:
            self.conv = conv_bn(in_channels, out_channels, kernel, stride, padding, dilation, groups)

    def emit(self, pb, x, out=None):
        conv, bn = (self.rep_conv, None) if hasattr(self, "rep_conv") else (self.conv.conv, self.conv.bn)
        if self.slope == SLOPE_SILU:
            # SiLU is not piecewise linear, so it cannot ride in the consumer's per-channel (scale, shift, slope) table:
            # the convolution keeps its pending BatchNorm and one elementwise combine stores silu(BN(conv(x))).
            return pb.ew([pb.conv(x, conv, bn, slope=1.0)], out_slope=SLOPE_SILU, out=out)
        return pb.conv(x, conv, bn, slope=self.slope, out=out)

    def switch_to_deploy(self):
        """repblocks.py:46-73.  (The reference builds `rep_conv` with out_channels=in_channels and then overwrites
        `.weight.data`; only the tensors matter, and those are identical here.)"""
        if hasattr(self, "rep_conv"):
            return
        self.rep_conv = fused_conv(self.conv.conv, [(self.conv.conv.weight, self.conv.bn)])
        del self.conv
        self.deploy = True
        self.__dict__.pop("_engine", None)


class RepBlock(PlanModule):
    """act(BN(conv_kxk(x)) + BN(conv_1x1(x)) [+ BN(x)])  -- repblocks.py:76-144 (ca_type None)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1,
                 padding_mode="zeros", deploy=False, ca_type=None, activation=nn.LeakyReLU, inplace=False,
                 identity=True):
        super().__init__()
        if ca_type not in (None, "none"):
            raise _lib.LhnError("RepBlock: attention inside the block is not built")
        self.deploy = deploy
        self.slope = act_slope(activation, inplace, positional=False)
        if deploy:
            self.rbr_reparam = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                         bias=True)
            return
        self.rbr_identity = (nn.BatchNorm2d(in_channels)
                             if identity and out_channels == in_channels and stride == 1 else None)
        self.rbr_dense = conv_bn(in_channels, out_channels, kernel_size, stride, padding, dilation, groups)
        self.rbr_1x1 = conv_bn(in_channels, out_channels, 1, stride, 0, 1, groups)

    def switch_to_deploy(self):
        """repblocks.py:169-236: k x k + centre-padded 1x1 + identity-as-convolution, each with its BN folded."""
        if hasattr(self, "rbr_reparam"):
            return
        branches = [(self.rbr_1x1.conv.weight, self.rbr_1x1.bn)]
        if self.rbr_identity is not None:
            branches.append((None, self.rbr_identity))
        branches.append((self.rbr_dense.conv.weight, self.rbr_dense.bn))
        self.rbr_reparam = fused_conv(self.rbr_dense.conv, branches)
        del self.rbr_dense, self.rbr_1x1
        if hasattr(self, "rbr_identity"):
            del self.rbr_identity
        self.deploy = True
        self.__dict__.pop("_engine", None)

    def emit(self, pb, x, out=None):
        if hasattr(self, "rbr_reparam"):
            return pb.conv(x, self.rbr_reparam, None, slope=self.slope, out=out)
        a = pb.conv(x, self.rbr_dense.conv, self.rbr_dense.bn, slope=1.0)
        b = pb.conv(x, self.rbr_1x1.conv, self.rbr_1x1.bn, slope=1.0)
        srcs = [a, b]
        if self.rbr_identity is not None:
            srcs.append(pb.bn_only(x, self.rbr_identity))
        return pb.ew(srcs, out_slope=self.slope, out=out)

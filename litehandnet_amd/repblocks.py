"""HIP-backed mirror of models/pose_estimation/liteHandNet/repblocks.py (train-time forward/backward).

Same constructor arguments and `state_dict()` keys as the reference classes; the arithmetic runs in
liblhn (csrc/k_conv_*.hip).  Deploy-time re-parameterisation (switch_to_deploy) is a "next" row."""
from torch import nn

from . import _lib
from .engine import PlanModule


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
    raise _lib.LhnError(f"activation {activation} is not expressible as a leaky slope (SiLU: not built yet)")


class RepConv(PlanModule):
    """act(BN(conv(x)))  -- repblocks.py:23-44."""

    def __init__(self, in_channels, out_channels, kernel=1, stride=1, padding=0, dilation=1, groups=1, deploy=False,
                 activation=nn.LeakyReLU, inplace=False):
        super().__init__()
        if deploy:
            raise _lib.LhnError("deploy-form RepConv is not built yet (SURVEY section 8f, rank 2)")
        self.deploy = False
        self.slope = act_slope(activation, inplace, positional=True)
        self.conv = conv_bn(in_channels, out_channels, kernel, stride, padding, dilation, groups)

    def emit(self, pb, x, out=None):
        return pb.conv(x, self.conv.conv, self.conv.bn, slope=self.slope, out=out)


class RepBlock(PlanModule):
    """act(BN(conv_kxk(x)) + BN(conv_1x1(x)) [+ BN(x)])  -- repblocks.py:76-144 (ca_type None)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1,
                 padding_mode="zeros", deploy=False, ca_type=None, activation=nn.LeakyReLU, inplace=False,
                 identity=True):
        super().__init__()
        if deploy or ca_type not in (None, "none"):
            raise _lib.LhnError("RepBlock: deploy form / attention inside the block are not built")
        self.deploy = False
        self.slope = act_slope(activation, inplace, positional=False)
        self.rbr_identity = (nn.BatchNorm2d(in_channels)
                             if identity and out_channels == in_channels and stride == 1 else None)
        self.rbr_dense = conv_bn(in_channels, out_channels, kernel_size, stride, padding, dilation, groups)
        self.rbr_1x1 = conv_bn(in_channels, out_channels, 1, stride, 0, 1, groups)

    def emit(self, pb, x, out=None):
        a = pb.conv(x, self.rbr_dense.conv, self.rbr_dense.bn, slope=1.0)
        b = pb.conv(x, self.rbr_1x1.conv, self.rbr_1x1.bn, slope=1.0)
        srcs = [a, b]
        if self.rbr_identity is not None:
            srcs.append(pb.bn_only(x, self.rbr_identity))
        return pb.ew(srcs, out_slope=self.slope, out=out)

"""HIP-backed mirror of models/pose_hg_ms_att.py -- the registered `mynet` (MultiScaleAttentionHourglass; SURVEY.md
section 8 row a13).  Same class names, constructor arguments, attribute names and Sequential indices as the reference,
hence the same state_dict keys (2,240,405 parameters).  The arithmetic is liblhn's: the convolutions are the kernels
of variants A/B, a biased conv in front of a BatchNorm keeps its bias in the BN finalize (lhn_bn_finalize conv_bias),
the BN -> SiLU -> conv unit is an identity-depthwise statistics pass + a SiLU combine, and the attention runs in
csrc/k_att.hip."""
from torch import nn

from . import _lib
from .engine import PlanModule
from .plan import SLOPE_SILU


def _slope_of(act):
    if isinstance(act, nn.ReLU):
        return 0.0
    if isinstance(act, nn.LeakyReLU):
        return float(act.negative_slope)
    raise _lib.LhnError(f"activation {act} after a convolution is not a leaky slope")


def emit_conv_bn_act(pb, x, seq, out=None):
    """Walk an nn.Sequential made of [Conv2d, BatchNorm2d, optional ReLU/LeakyReLU] groups (the reference writes its
    blocks that way) and emit one fused conv+BN(+activation) per group.  `out` receives the last group's output."""
    mods = list(seq)
    groups, i = [], 0
    while i < len(mods):
        conv, bn = mods[i], mods[i + 1]
        assert isinstance(conv, nn.Conv2d) and isinstance(bn, nn.modules.batchnorm._BatchNorm), (type(conv), type(bn))
        i += 2
        slope = 1.0
        if i < len(mods) and not isinstance(mods[i], nn.Conv2d):
            slope = _slope_of(mods[i])
            i += 1
        groups.append((conv, bn, slope))
    for j, (conv, bn, slope) in enumerate(groups):
        x = pb.conv(x, conv, bn, slope=slope, out=out if j == len(groups) - 1 else None)
    return x


def _conv_bn(cin, cout, k, stride=1, pad=0, groups=1, bias=True, act=None, dil=1):
    mods = [nn.Conv2d(cin, cout, k, stride, pad, dil, groups, bias=bias), nn.BatchNorm2d(cout)]
    return mods + ([act] if act is not None else [])


class DWConv(PlanModule):
    """pose_hg_ms_att.py:7-22."""

    def __init__(self, in_channel, out_channel, stride=1, padding=1, dilation=1, mid_relu=True, last_relu=True, bias=False):
        super().__init__()
        self.depthwise_conv = nn.Sequential(*_conv_bn(in_channel, in_channel, 3, stride, padding, in_channel, bias,
                                                      dil=dilation))
        self.mid_relu = nn.ReLU() if mid_relu else nn.Identity()
        self.pointwise_conv = nn.Sequential(*_conv_bn(in_channel, out_channel, 1, bias=bias))
        self.last_relu = nn.ReLU() if last_relu else nn.Identity()

    def emit(self, pb, x, out=None):
        s1 = 0.0 if isinstance(self.mid_relu, nn.ReLU) else 1.0
        s2 = 0.0 if isinstance(self.last_relu, nn.ReLU) else 1.0
        t = pb.conv(x, self.depthwise_conv[0], self.depthwise_conv[1], slope=s1)
        return pb.conv(t, self.pointwise_conv[0], self.pointwise_conv[1], slope=s2, out=out)


class BottleNeck(PlanModule):
    """pose_hg_ms_att.py:24-39: relu(x + 1x1 -> 3x3 -> 1x1), biased convs + BN."""

    def __init__(self, channel):
        super().__init__()
        q = channel // 4
        self.conv = nn.Sequential(*_conv_bn(channel, q, 1, act=nn.ReLU(inplace=True)),
                                  *_conv_bn(q, q, 3, 1, 1, act=nn.ReLU(inplace=True)), *_conv_bn(q, channel, 1))

    def emit(self, pb, x, out=None):
        return pb.ew([x, emit_conv_bn_act(pb, x, self.conv)], out_slope=0.0, out=out)


class BasicBlock(PlanModule):
    """pose_hg_ms_att.py:42-61."""

    def __init__(self, inp_dim, out_dim, stride=1):
        super().__init__()
        self.conv = nn.Sequential(*_conv_bn(inp_dim, out_dim, 3, stride, 1, act=nn.ReLU(inplace=True)),
                                  *_conv_bn(out_dim, out_dim, 3, 1, 1))
        if stride == 2 or inp_dim != out_dim:
            self.skip_layer = nn.Sequential(*_conv_bn(inp_dim, out_dim, 1, stride, 0))
        else:
            self.skip_layer = nn.Identity()

    def emit(self, pb, x, out=None):
        t = emit_conv_bn_act(pb, x, self.conv)
        s = emit_conv_bn_act(pb, x, self.skip_layer) if isinstance(self.skip_layer, nn.Sequential) else x
        return pb.ew([s, t], out_slope=0.0, out=out)


class Residual(PlanModule):
    """pose_hg_ms_att.py:63-72."""

    def __init__(self, inp_dim, out_dim, stride=1, num_block=2):
        super().__init__()
        self.conv1 = BasicBlock(inp_dim, out_dim, stride)
        self.blocks = nn.Sequential(*[BottleNeck(out_dim) for _ in range(num_block)])

    def emit(self, pb, x, out=None):
        x = self.conv1.emit(pb, x)
        for b in self.blocks:
            x = b.emit(pb, x)
        return x


class BRC(PlanModule):
    """pose_hg_ms_att.py:74-90: BatchNorm -> SiLU -> conv (the class is named BN+ReLU+Conv, the code uses SiLU)."""

    def __init__(self, inp_dim, out_dim, kernel_size=3, stride=1, padding=1, bias=False, dilation=1):
        super().__init__()
        self.inp_dim = inp_dim
        self.conv = nn.Conv2d(inp_dim, out_dim, kernel_size, stride, padding=padding, bias=bias, dilation=dilation)
        self.silu = nn.SiLU(inplace=True)
        self.bn = nn.BatchNorm2d(inp_dim)

    def emit(self, pb, x, out=None):
        t = pb.bn_only(x, self.bn)                       # statistics pass; normalisation stays pending
        u = pb.ew([t], out_slope=SLOPE_SILU)             # SiLU of the normalised value, stored plain
        return pb.conv(u, self.conv, None, out=out)


class ME_att(PlanModule):
    """pose_hg_ms_att.py:141-193."""

    def __init__(self, in_c, out_c, p_drop=0.3):
        super().__init__()
        m = in_c // 2
        self.conv1 = BRC(in_c, m, 1, 1, 0)
        self.mid1_conv = nn.ModuleList([nn.Sequential(DWConv(m, m // 2), DWConv(m // 2, m // 2)),
                                        nn.Sequential(DWConv(m, m), DWConv(m, m))])
        self.mid2_conv = nn.ModuleList([nn.Sequential(DWConv(m, m // 2, dilation=2, padding=2), DWConv(m // 2, m // 2)),
                                        nn.Sequential(DWConv(m, m, dilation=2, padding=2), DWConv(m, m))])
        self.conv2 = BRC(in_c, out_c, 1, 1, 0, bias=False)
        self.att = nn.Sequential(nn.AdaptiveAvgPool2d((3, 3)), nn.BatchNorm2d(out_c), nn.ReLU(),
                                 nn.Conv2d(out_c, out_c, 3, 1, 0, groups=out_c), nn.Flatten(), nn.Dropout(p=p_drop),
                                 nn.Linear(out_c, out_c), nn.Sigmoid())
        self.mid_c = m

    def emit(self, pb, x, out=None):
        m = self.conv1.emit(pb, x)
        for r in range(2):
            half = self.mid_c // 2 if r == 0 else self.mid_c
            cat = pb.new(m.H, m.W, 2 * half)
            for j, branch in enumerate((self.mid1_conv[r], self.mid2_conv[r])):
                branch[1].emit(pb, branch[0].emit(pb, m), out=pb.slice(cat, j * half, half))
            m = cat
        y = self.conv2.emit(pb, pb.ew([m, x]))
        return pb.me_attention(y, self.att)


class EncoderDecoder(PlanModule):
    """pose_hg_ms_att.py:93-138."""

    def __init__(self, num_levels=5, inp_dim=128, num_blocks=[], p_drop=0.3):
        super().__init__()
        assert len(num_blocks) == num_levels - 1
        self.num_levels = num_levels
        self.encoder, self.decoder = nn.ModuleList([]), nn.ModuleList([])
        self.encoder.append(ME_att(inp_dim, inp_dim, p_drop))
        for i in range(num_levels - 1):
            self.encoder.append(Residual(inp_dim, inp_dim, 2, num_blocks[i]))
            self.decoder.append(Residual(inp_dim, inp_dim))
        self.decoder.append(ME_att(inp_dim, inp_dim, p_drop))

    def emit(self, pb, x, out=None):
        enc = []
        for layer in self.encoder:
            x = layer.emit(pb, x)
            enc.append(x)
        short = pb.avgpool(enc[0], enc[-1].H, enc[-1].W)
        for i, layer in enumerate(self.decoder):
            peer = enc[self.num_levels - 1 - i]
            x = pb.ew([layer.emit(pb, peer if i == 0 else x), short if i == 0 else peer])   # (+ nearest upsample)
        return x


class my_pelee_stem(PlanModule):
    """pose_hg_ms_att.py:196-228."""
    consumes_image = True

    def __init__(self, out_channel=256, min_mid_c=32):
        super().__init__()
        m = out_channel // 4 if out_channel // 4 >= min_mid_c else min_mid_c
        self.conv1 = nn.Sequential(*_conv_bn(3, m, 3, 2, 1, bias=False, act=nn.LeakyReLU(inplace=True)),
                                   *_conv_bn(m, m, 3, 1, 1, m, False, act=nn.LeakyReLU(inplace=True)))
        self.branch1 = nn.Sequential(*_conv_bn(m, m, 1, act=nn.ReLU(True)), *_conv_bn(m, m, 3, 2, 1, act=nn.ReLU(True)))
        self.branch2 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.conv1x1 = nn.Conv2d(m * 2, out_channel, 1, 1, 0)
        self.mid = m

    def emit(self, pb, x, out=None):
        m = self.mid
        t = emit_conv_bn_act(pb, x, self.conv1)
        cat = pb.new((t.H + 1) // 2, (t.W + 1) // 2, 2 * m)
        emit_conv_bn_act(pb, t, self.branch1, out=pb.slice(cat, 0, m))
        pb.maxpool(t, out=pb.slice(cat, m, m))
        return pb.conv(cat, self.conv1x1, None, out=out)


class MultiScaleAttentionHourglass(PlanModule):
    """pose_hg_ms_att.py:231-265.  cfg.MODEL keys: num_stage, input_channel, output_channel, num_block,
    output_acitivation (the reference's spelling)."""
    consumes_image = True

    def __init__(self, cfg):
        super().__init__()
        M = cfg.MODEL
        num_stage = M.get("num_stage", 4)
        inp_dim = M.get("input_channel", 128)
        oup_dim = M.get("output_channel", cfg.DATASET.num_joints)
        num_block = M.get("num_block", [2, 2, 2])
        self.with_activation = M.get("output_acitivation", False)
        self.p_drop = float(M.get("ca_dropout", 0.3))
        self.pre = my_pelee_stem(inp_dim)
        self.hgs = EncoderDecoder(num_stage, inp_dim, num_block, p_drop=self.p_drop)
        self.features = nn.Sequential(BottleNeck(inp_dim), nn.Conv2d(inp_dim, inp_dim, 1, 1, 0), nn.BatchNorm2d(inp_dim),
                                      nn.LeakyReLU())
        self.outs = nn.Conv2d(inp_dim, oup_dim, 1, 1, 0)
        self.init_weights()

    def emit(self, pb, x, out=None):
        y = self.hgs.emit(pb, self.pre.emit(pb, x))
        y = self.features[0].emit(pb, y)
        y = pb.conv(y, self.features[1], self.features[2], slope=_slope_of(self.features[3]))
        return pb.conv(y, self.outs, None, nchw_out=True)

    def forward(self, imgs):
        preds = super().forward(imgs)
        if self.with_activation:
            # pose_hg_ms_att.py:251-252.  No shipped config can switch this on (they spell the key `output_activation` /
            # `output_swish`, the model reads `output_acitivation`), so it stays one library elementwise op on the head.
            preds = nn.functional.leaky_relu(preds, 0.5)
        return preds

    def init_weights(self):
        # pose_hg_ms_att.py:256-262: conv weight ~ N(0,1), bias 0 (weight_init.py:28-32); BatchNorm gamma 1, beta 0
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

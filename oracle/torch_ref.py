"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in plain PyTorch fp32 / NCHW, of the reference's convolutional
heatmap path.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import this file.  It is pinned against the real
reference by `tests/golden/make_golden.py` (run in the build container, where
`/root/reference` is importable) and the committed fixtures in `tests/golden/`.

Every class keeps the reference's attribute names because the `state_dict()`
keys are the compatibility contract (checkpoint row); the
bodies are written from the behaviour, not copied.

Reference behaviour followed (paths relative to the reference tree):
  * conv+BN(+act) unit            models/pose_estimation/liteHandNet/repblocks.py:8-44
  * 3-branch re-param block        models/pose_estimation/liteHandNet/repblocks.py:76-144
  * channel attention              models/pose_estimation/liteHandNet/common.py:40-66
  * variant A (registered name)    models/pose_estimation/liteHandNet/liteHandNet.py:8-238
  * variant B (MSRB hourglass)     models/pose_estimation/liteHandNet/litehourglass.py:13-237
  * `mynet` (a13)                  models/pose_hg_ms_att.py:7-265
  * deploy-time re-parameterisation repblocks.py:46-73,169-236, common.py:68-90, liteHandNet.py:240-244
  * init                           models/weight_init.py:21-32
  * loss                           loss/loss.py:69-114, loss/heatmapLoss.py:228-265
"""
import torch
from torch import nn
import torch.nn.functional as F

_ACTS = {"leakyrelu": nn.LeakyReLU, "relu": nn.ReLU, "silu": nn.SiLU}


def _act(kind):
    """kind: None | class | str -> module instance."""
    if kind is None:
        return nn.Identity()
    if isinstance(kind, str):
        kind = _ACTS[kind.lower()]
    return kind()


def _repconv_act(kind, inplace):
    """repblocks.py:29-30 builds `activation(inplace)` POSITIONALLY.  For nn.LeakyReLU the first
    positional parameter is negative_slope, so the unit's non-linearity is LeakyReLU(slope=float(inplace)):
    ReLU-like (slope 0) when inplace=False, the identity (slope 1) when inplace=True.  nn.ReLU / nn.SiLU
    take `inplace` first and behave normally.  This is the reference's behaviour, so it is the oracle's."""
    if kind is None:
        return nn.Identity()
    if isinstance(kind, str):
        kind = _ACTS[kind.lower()]
    if kind is nn.LeakyReLU:
        return nn.LeakyReLU(negative_slope=float(inplace))
    return kind()


def _conv_bn(cin, cout, k, stride, pad, dil=1, groups=1):
    seq = nn.Sequential()
    seq.add_module("conv", nn.Conv2d(cin, cout, k, stride, pad, dil, groups, bias=False))
    seq.add_module("bn", nn.BatchNorm2d(cout))
    return seq


def _fold_bn(kernel, bn):
    """BN(conv_k(x)) in eval mode == conv_{k*t}(x) + (beta - mean*gamma/std), t = gamma/std per output channel
    (repblocks.py:49-58, 197-221).  Same operation order as the reference (sqrt, divide, multiply)."""
    std = (bn.running_var + bn.eps).sqrt()
    t = (bn.weight / std).reshape(-1, 1, 1, 1)
    return kernel * t, bn.bias - bn.running_mean * bn.weight / std


def fold_bn_np(kernel, gamma, beta, mean, var, eps):
    """_fold_bn in numpy float32: every operation correctly rounded (IEEE-754).  torch's CPU `sqrt` (MKL VML) is off by
    one ulp on a few inputs, so the HIP kernel -- whose sqrt/divide are correctly rounded -- is pinned bit-for-bit against
    THIS function and within 2 ulp against the reference-generated fixture."""
    import numpy as np
    f = np.float32
    std = np.sqrt(var.astype(f) + f(eps))
    t = gamma.astype(f) / std
    return kernel.astype(f) * t.reshape(-1, 1, 1, 1), beta.astype(f) - mean.astype(f) * gamma.astype(f) / std


def _swap_in(mod, name, like, kernel, bias, drop):
    """Install `name` = biased conv shaped like `like` holding (kernel, bias); remove the train-time branches."""
    conv = nn.Conv2d(like.in_channels, kernel.shape[0], like.kernel_size, like.stride, like.padding, like.dilation,
                     like.groups, bias=True)
    conv.weight.data = kernel.detach()
    conv.bias.data = bias.detach()
    setattr(mod, name, conv)
    for p in mod.parameters():
        p.detach_()
    for d in drop:
        if hasattr(mod, d):
            delattr(mod, d)
    mod.deploy = True


def deploy_model(model):
    """liteHandNet.py:240-244 / litehourglass.py:233-237: every module that can re-parameterise does."""
    for m in model.modules():
        if hasattr(m, "switch_to_deploy"):
            m.switch_to_deploy()
    model.deploy = True
    return model


class RepConv(nn.Module):
    """act(BN(conv(x))); keys `conv.conv.weight`, `conv.bn.*`; after deploy `rep_conv.{weight,bias}`."""

    def __init__(self, cin, cout, kernel=1, stride=1, padding=0, dilation=1, groups=1,
                 activation=nn.LeakyReLU, inplace=False):
        super().__init__()
        self.conv = _conv_bn(cin, cout, kernel, stride, padding, dilation, groups)
        self.nonlinearity = _repconv_act(activation, inplace)

    def forward(self, x):
        if hasattr(self, "rep_conv"):
            return self.nonlinearity(self.rep_conv(x))
        return self.nonlinearity(self.conv(x))

    def switch_to_deploy(self):  # repblocks.py:46-73
        if hasattr(self, "rep_conv"):
            return
        k, b = _fold_bn(self.conv.conv.weight, self.conv.bn)
        _swap_in(self, "rep_conv", self.conv.conv, k, b, ("conv",))


class RepBlock(nn.Module):
    """act(BN(conv_kxk(x)) + BN(conv_1x1(x)) [+ BN(x)])."""

    def __init__(self, cin, cout, k=3, stride=1, padding=1, groups=1, activation=nn.LeakyReLU):
        super().__init__()
        self.rbr_identity = nn.BatchNorm2d(cin) if (cin == cout and stride == 1) else None
        self.rbr_dense = _conv_bn(cin, cout, k, stride, padding, 1, groups)
        self.rbr_1x1 = _conv_bn(cin, cout, 1, stride, 0, 1, groups)
        self.nonlinearity = _act(activation)

    def forward(self, x):
        if hasattr(self, "rbr_reparam"):
            return self.nonlinearity(self.rbr_reparam(x))
        y = self.rbr_dense(x) + self.rbr_1x1(x)
        if self.rbr_identity is not None:
            y = y + self.rbr_identity(x)
        return self.nonlinearity(y)

    def switch_to_deploy(self):  # repblocks.py:169-236: k x k  +  centre-padded 1x1  +  identity-as-conv
        if hasattr(self, "rbr_reparam"):
            return
        dense = self.rbr_dense.conv
        kk, bk = _fold_bn(dense.weight, self.rbr_dense.bn)
        k1, b1 = _fold_bn(self.rbr_1x1.conv.weight, self.rbr_1x1.bn)
        pad = dense.kernel_size[0] // 2
        rest_k, rest_b = F.pad(k1, [pad] * 4), b1
        if self.rbr_identity is not None:
            cin, cin_g = dense.in_channels, dense.in_channels // dense.groups
            eye = torch.zeros(cin, cin_g, *dense.kernel_size)
            eye[torch.arange(cin), torch.arange(cin) % cin_g, pad, pad] = 1
            ki, bi = _fold_bn(eye.to(dense.weight.device), self.rbr_identity)
            rest_k, rest_b = rest_k + ki, rest_b + bi
        else:                                   # the reference adds the integer 0 for a missing branch
            rest_k, rest_b = rest_k + 0, rest_b + 0
        _swap_in(self, "rbr_reparam", dense, kk + rest_k, bk + rest_b, ("rbr_dense", "rbr_1x1", "rbr_identity"))


class ChannelAttension(nn.Module):
    """x * sigmoid(W2 lrelu(W1 drop(BN(dw3x3_valid(avgpool_3x3(x))))))."""

    def __init__(self, c, p_drop=0.3):
        super().__init__()
        self.conv3x3 = nn.Sequential()
        self.conv3x3.add_module("conv", nn.Conv2d(c, c, 3, 1, 0, groups=c, bias=False))
        self.conv3x3.add_module("bn", nn.BatchNorm2d(c))
        self.conv1x1 = nn.Sequential(
            nn.Dropout2d(p=p_drop),
            nn.Conv2d(c, c // 2, 1),
            nn.LeakyReLU(),
            nn.Conv2d(c // 2, c, 1),
            nn.Sigmoid())

    def forward(self, x):
        pooled = F.adaptive_avg_pool2d(x, (3, 3))
        att = self.rbr_reparam(pooled) if hasattr(self, "rbr_reparam") else self.conv3x3(pooled)
        return x * self.conv1x1(att)

    def switch_to_deploy(self):  # common.py:68-90
        if hasattr(self, "rbr_reparam"):
            return
        k, b = _fold_bn(self.conv3x3.conv.weight, self.conv3x3.bn)
        _swap_in(self, "rbr_reparam", self.conv3x3.conv, k, b, ("conv3x3",))


class FixedMask(nn.Module):
    """Test helper: stands in for nn.Dropout2d / nn.Dropout with a GIVEN mask (values 0 or 1/keep, one per (n, c)), so that
    the HIP path and the oracle can be compared with dropout switched on (common.py:57, pose_hg_ms_att.py:171)."""

    def __init__(self, mask):
        super().__init__()
        self.register_buffer("mask", mask.clone())

    def forward(self, x):
        return x * self.mask.view(self.mask.shape + (1,) * (x.dim() - 2)).to(x.dtype)


def install_masks(model, masks):
    """masks: {module name: [N, C] tensor}; replaces the dropout of every named attention module.  Returns model."""
    mods = dict(model.named_modules())
    for name, mk in masks.items():
        m = mods[name]
        if isinstance(m, ChannelAttension):
            m.conv1x1[0] = FixedMask(mk)
        elif isinstance(m, ME_att):
            m.att[5] = FixedMask(mk)
        elif isinstance(m, nn.Sequential) and len(m) > 5 and isinstance(m[5], (nn.Dropout, FixedMask)):
            m[5] = FixedMask(mk)            # ME_att.att addressed directly (pose_hg_ms_att.py:165-174)
        else:
            raise TypeError(f"{name}: {type(m).__name__} has no dropout")
    return model


class SEBlock(nn.Module):
    """common.py:23-37: x * sigmoid(up(relu(down(avg_pool2d(x, kernel = width)))))."""

    def __init__(self, c, internal):
        super().__init__()
        self.down = nn.Conv2d(c, internal, 1)
        self.up = nn.Conv2d(internal, c, 1)
        self.input_channels = c

    def forward(self, x):
        g = torch.sigmoid(self.up(F.relu(self.down(F.avg_pool2d(x, kernel_size=x.size(3))))))
        return x * g.view(-1, self.input_channels, 1, 1)


def _make_ca(kind, c, p_drop):
    if kind == "ca":
        return ChannelAttension(c, p_drop)
    if kind == "se":
        return SEBlock(c, c // 16)
    if kind == "none":
        return nn.Identity()
    raise ValueError(f"ca_type {kind!r}: the hot path covers 'ca' and 'none' only")


# --------------------------------------------------------------------------
# Variant B: MSRB hourglass  (litehourglass.py)
# --------------------------------------------------------------------------
class MSRB(nn.Module):
    def __init__(self, cin, cout, ca_type="none", p_drop=0.3):
        super().__init__()
        h = cin // 2
        self.branch1 = nn.ModuleList([RepConv(h, h, 3, 1, 1, groups=h, activation=None) for _ in range(2)])
        self.branch2 = nn.ModuleList([RepConv(h, h, 3, 1, 2, 2, groups=h, activation=None) for _ in range(2)])
        self.ca = nn.ModuleList([_make_ca(ca_type, cout, p_drop) for _ in range(2)])
        self.conv = RepConv(cin, cout, 1, 1, 0)

    def forward(self, x):
        acc = x
        for b1, b2, ca in zip(self.branch1, self.branch2, self.ca):
            lo, hi = torch.chunk(acc, 2, dim=1)
            acc = acc + ca(torch.cat([b1(lo), b2(hi)], dim=1))
        return self.conv(acc + x)


class RepBasicUnit(nn.Module):
    def __init__(self, cin, cout, ca_type="ca", p_drop=0.3):
        super().__init__()
        self.left_part = cin // 2
        r_in, r_out = cin - self.left_part, cout - self.left_part
        self.conv = nn.Sequential(RepConv(r_in, r_out, 1),
                                  RepConv(r_out, r_out, 3, padding=1, groups=r_out))
        self.ca = _make_ca(ca_type, cout, p_drop)

    def forward(self, x):
        keep, work = x[:, :self.left_part], x[:, self.left_part:]
        return self.ca(torch.cat([keep, self.conv(work)], dim=1))


class _HourglassB(nn.Module):
    def __init__(self, num_stage, c, msrb_ca, rbu_ca, p_drop):
        super().__init__()
        self.num_stage = num_stage
        self.encoder, self.decoder = nn.ModuleList(), nn.ModuleList()
        self.maxpool = nn.MaxPool2d(2, 2)
        for i in range(num_stage):
            for lst in (self.encoder, self.decoder):
                first = MSRB(c, c, msrb_ca, p_drop) if i == 0 else RepBasicUnit(c, c, rbu_ca, p_drop)
                lst.append(nn.Sequential(first, RepBasicUnit(c, c, rbu_ca, p_drop)))

    def forward(self, x):
        skips = []
        for i, enc in enumerate(self.encoder):
            x = enc(x)
            skips.append(x)
            if i + 1 < self.num_stage:
                x = self.maxpool(x)
        outs = []
        for i in reversed(range(self.num_stage)):
            if i == self.num_stage - 1:
                x = self.decoder[i](skips[i])
                x = x + F.adaptive_avg_pool2d(skips[0], skips[-1].shape[2:])
            else:
                x = F.interpolate(x, size=skips[i].shape[2:]) + skips[i]
                x = self.decoder[i](x)
            outs.append(x)
        return tuple(outs)


class _StemB(nn.Module):
    def __init__(self, c, p_drop):
        super().__init__()
        m = max(c // 4, 32)
        self.conv1 = nn.Sequential(RepConv(3, m, 3, 2, 1), RepConv(m, m, 3, 1, 1, groups=m))
        self.branch1 = nn.Sequential(RepConv(m, m, 1), RepConv(m, m, 3, 2, 1, groups=m, activation=None),
                                     RepConv(m, m, 1))
        self.branch2 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.conv2 = nn.Sequential(RepConv(2 * m, c), RepBasicUnit(c, c, "ca", p_drop),
                                   RepBasicUnit(c, c, "ca", p_drop))

    def forward(self, x):
        t = self.conv1(x)
        return self.conv2(torch.cat([self.branch1(t), self.branch2(t)], dim=1))


class LiteHourglassNet(nn.Module):
    """Variant B.  cfg keys: MODEL.{num_stage,msrb_ca,rbu_ca,input_channel,output_channel}."""

    def __init__(self, cfg, p_drop=0.3):
        super().__init__()
        M = cfg.MODEL
        c = M.get("input_channel", 256)
        self.stem = _StemB(c, p_drop)
        self.backone = _HourglassB(M.get("num_stage", 4), c, M.get("msrb_ca", "ca"), M.get("rbu_ca", "ca"), p_drop)
        self.neck = nn.Sequential(RepBasicUnit(c, c, "ca", p_drop), RepBasicUnit(c, c, "ca", p_drop))
        self.head = nn.Conv2d(c, M.get("output_channel", cfg.DATASET.num_joints), 1)
        self.init_weights()

    def forward(self, x):
        return self.head(self.neck(self.backone(self.stem(x))[-1]))

    def init_weights(self):  # litehourglass.py:224-230: conv ~ N(0,1), bias 0; BN gamma 1, beta 0
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 1)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)


# --------------------------------------------------------------------------
# Variant A: registered `litehandnet`  (liteHandNet.py)
# --------------------------------------------------------------------------
class DWConv(nn.Module):
    def __init__(self, cin, cout, padding=1, dilation=1, activation=nn.LeakyReLU):
        super().__init__()
        self.depthwise_conv = RepConv(cin, cin, 3, 1, padding, dilation, groups=cin, activation=activation)
        self.pointwise_conv = RepConv(cin, cout, 1, activation=activation)

    def forward(self, x):
        return self.pointwise_conv(self.depthwise_conv(x))


class BottleNeck(nn.Module):
    def __init__(self, c, reduction=4, activation=nn.LeakyReLU):
        super().__init__()
        m = c // reduction
        self.conv = nn.Sequential(RepConv(c, m, 1, activation=activation, inplace=True),
                                  RepConv(m, m, 3, 1, 1, activation=activation, inplace=True),
                                  RepConv(m, c, 1, activation=None))
        self.activation = _act(activation)

    def forward(self, x):
        return self.activation(x + self.conv(x))


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride=1, activation=nn.LeakyReLU):
        super().__init__()
        self.conv = nn.Sequential(RepConv(cin, cout, 3, stride, 1, activation=activation, inplace=True),
                                  RepConv(cin, cout, 3, 1, 1, activation=None))
        self.skip_layer = (RepConv(cin, cout, 1, stride, 0, activation=None)
                           if (stride == 2 or cin != cout) else nn.Identity())
        self.activation = _act(activation)

    def forward(self, x):
        return self.activation(self.skip_layer(x) + self.conv(x))


class Residual(nn.Module):
    def __init__(self, cin, cout, stride, num_block, reduction, activation):
        super().__init__()
        self.conv1 = BasicBlock(cin, cout, stride, activation)
        self.blocks = nn.Sequential(*[BottleNeck(cout, reduction, activation) for _ in range(num_block)])

    def forward(self, x):
        return self.blocks(self.conv1(x))


class MSAB(nn.Module):
    def __init__(self, cin, cout, ca_type="ca", activation=nn.LeakyReLU, p_drop=0.3):
        super().__init__()
        m = cin // 2
        a = activation
        self.conv1 = RepConv(cin, m, 1, activation=a, inplace=True)
        self.mid1_conv = nn.ModuleList([
            nn.Sequential(DWConv(m, m // 2, activation=a), DWConv(m // 2, m // 2, activation=a)),
            nn.Sequential(DWConv(m, m, activation=a), DWConv(m, m, activation=a))])
        self.mid2_conv = nn.ModuleList([
            nn.Sequential(DWConv(m, m // 2, 2, 2, activation=a), DWConv(m // 2, m // 2, activation=a)),
            nn.Sequential(DWConv(m, m, 2, 2, activation=a), DWConv(m, m, activation=a))])
        self.conv2 = RepConv(cin, cout, 1, activation=a, inplace=True)
        self.ca = _make_ca(ca_type, cout, p_drop)

    def forward(self, x):
        t = self.conv1(x)
        for r in range(2):
            t = torch.cat([self.mid1_conv[r](t), self.mid2_conv[r](t)], dim=1)
        return self.ca(self.conv2(t + x))


class _HourglassA(nn.Module):
    def __init__(self, levels, c, blocks, ca_type, reduction, act, p_drop):
        super().__init__()
        assert len(blocks) == levels - 1
        self.num_levels = levels
        self.encoder = nn.ModuleList([MSAB(c, c, ca_type, p_drop=p_drop)])
        self.decoder = nn.ModuleList()
        for nb in blocks:
            self.encoder.append(Residual(c, c, 2, nb, reduction, act))
            self.decoder.append(Residual(c, c, 1, nb, reduction, act))
        self.decoder.append(MSAB(c, c, ca_type, p_drop=p_drop))

    def forward(self, x):
        enc = []
        for layer in self.encoder:
            x = layer(x)
            enc.append(x)
        short = F.adaptive_avg_pool2d(enc[0], enc[-1].shape[2:])
        outs = []
        for i, layer in enumerate(self.decoder):
            peer = enc[self.num_levels - 1 - i]
            if i == 0:
                x = layer(peer) + short
            else:
                x = F.interpolate(layer(x), size=peer.shape[2:]) + peer
            outs.append(x)
        return tuple(outs)


class _StemA(nn.Module):
    def __init__(self, cout, act, min_mid=32):
        super().__init__()
        m = max(cout // 4, min_mid)
        self.conv1 = nn.Sequential(RepBlock(3, m, 3, 2, 1, activation=act),
                                   RepBlock(m, m, 7, 1, 3, groups=m, activation=act))
        self.branch1 = nn.Sequential(RepConv(m, m, 1, activation=act, inplace=True),
                                     RepConv(m, m, 3, 2, 1, activation=act, inplace=True))
        self.branch2 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.conv1x1 = nn.Conv2d(2 * m, cout, 1)

    def forward(self, x):
        t = self.conv1(x)
        return self.conv1x1(torch.cat([self.branch1(t), self.branch2(t)], dim=1))


class LiteHandNet(nn.Module):
    """Variant A.  cfg keys: MODEL.{num_stage,input_channel,output_channel,num_block,ca_type,reduction,activation}."""

    def __init__(self, cfg, p_drop=0.3):
        super().__init__()
        M = cfg.MODEL
        c = M.get("input_channel", 128)
        red = M.get("reduction", 2)
        ca = M.get("ca_type", "ca")
        act = _ACTS[M.get("activation", "LeakyReLU").lower()]
        assert red in (2, 4)
        self.pre = _StemA(c, act)
        self.hgs = _HourglassA(M.get("num_stage", 4), c, M.get("num_block", [2, 2, 2]), ca, red, act, p_drop)
        self.features = nn.Sequential(BottleNeck(c, 2, act), RepConv(c, c, 1, activation=act, inplace=True))
        self.out_layer = nn.Conv2d(c, M.get("output_channel", cfg.DATASET.num_joints), 1)
        self.init_weights()

    def forward(self, x):
        return self.out_layer(self.features(self.hgs(self.pre(x))[-1]))

    def init_weights(self):  # liteHandNet.py:236-238 + weight_init.py:28-32: EVERY .weight ~ N(0,1) incl. BN gamma
        for m in self.modules():
            w = getattr(m, "weight", None)
            if isinstance(w, torch.Tensor):
                nn.init.normal_(w, 0, 1)
            b = getattr(m, "bias", None)
            if isinstance(b, torch.Tensor):
                nn.init.zeros_(b)



# --------------------------------------------------------------------------
# `mynet`: MultiScaleAttentionHourglass  (models/pose_hg_ms_att.py:7-265)
# Same topology as variant A built from plain Conv2d(+bias)+BN+ReLU, a BN->SiLU->conv pre-activation unit and a
# pooled-BN-ReLU-dw3x3-Linear attention.  Attribute names / Sequential indices follow the reference's state_dict.
# --------------------------------------------------------------------------
def _seq(*mods):
    return nn.Sequential(*mods)


def _cbr(cin, cout, k, stride=1, pad=0, groups=1, bias=True, act=None, dil=1):
    """[conv, BN] (+ activation module when given) as a flat list -- callers splice it into a Sequential."""
    mods = [nn.Conv2d(cin, cout, k, stride, pad, dil, groups, bias=bias), nn.BatchNorm2d(cout)]
    if act is not None:
        mods.append(act)
    return mods


class MyDWConv(nn.Module):
    """pose_hg_ms_att.py:7-22: dw3x3+BN, ReLU, 1x1+BN, ReLU (bias=False)."""

    def __init__(self, cin, cout, padding=1, dilation=1):
        super().__init__()
        self.depthwise_conv = _seq(*_cbr(cin, cin, 3, 1, padding, cin, False, dil=dilation))
        self.mid_relu = nn.ReLU()
        self.pointwise_conv = _seq(*_cbr(cin, cout, 1, bias=False))
        self.last_relu = nn.ReLU()

    def forward(self, x):
        return self.last_relu(self.pointwise_conv(self.mid_relu(self.depthwise_conv(x))))


class MyBottleNeck(nn.Module):
    """:24-39: relu(x + [1x1 C->C/4, 3x3, 1x1 ->C]) with biased convs + BN."""

    def __init__(self, c):
        super().__init__()
        q = c // 4
        self.conv = _seq(*_cbr(c, q, 1, act=nn.ReLU()), *_cbr(q, q, 3, 1, 1, act=nn.ReLU()), *_cbr(q, c, 1))

    def forward(self, x):
        return F.relu(x + self.conv(x))


class MyBasicBlock(nn.Module):
    """:42-61."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv = _seq(*_cbr(cin, cout, 3, stride, 1, act=nn.ReLU()), *_cbr(cout, cout, 3, 1, 1))
        self.skip_layer = _seq(*_cbr(cin, cout, 1, stride)) if (stride == 2 or cin != cout) else nn.Identity()

    def forward(self, x):
        return F.relu(self.skip_layer(x) + self.conv(x))


class MyResidual(nn.Module):
    """:63-72."""

    def __init__(self, cin, cout, stride=1, num_block=2):
        super().__init__()
        self.conv1 = MyBasicBlock(cin, cout, stride)
        self.blocks = _seq(*[MyBottleNeck(cout) for _ in range(num_block)])

    def forward(self, x):
        return self.blocks(self.conv1(x))


class BRC(nn.Module):
    """:74-90: BatchNorm -> SiLU -> conv (registration order conv, silu, bn)."""

    def __init__(self, cin, cout, k=3, stride=1, padding=1, bias=False, dilation=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, padding, dilation, bias=bias)
        self.silu = nn.SiLU()
        self.bn = nn.BatchNorm2d(cin)

    def forward(self, x):
        return self.conv(self.silu(self.bn(x)))


class ME_att(nn.Module):
    """:141-193."""

    def __init__(self, cin, cout, p_drop=0.3):
        super().__init__()
        m = cin // 2
        self.conv1 = BRC(cin, m, 1, 1, 0)
        self.mid1_conv = nn.ModuleList([_seq(MyDWConv(m, m // 2), MyDWConv(m // 2, m // 2)),
                                        _seq(MyDWConv(m, m), MyDWConv(m, m))])
        self.mid2_conv = nn.ModuleList([_seq(MyDWConv(m, m // 2, 2, 2), MyDWConv(m // 2, m // 2)),
                                        _seq(MyDWConv(m, m, 2, 2), MyDWConv(m, m))])
        self.conv2 = BRC(cin, cout, 1, 1, 0)
        self.att = _seq(nn.AdaptiveAvgPool2d((3, 3)), nn.BatchNorm2d(cout), nn.ReLU(),
                        nn.Conv2d(cout, cout, 3, 1, 0, groups=cout), nn.Flatten(), nn.Dropout(p=p_drop),
                        nn.Linear(cout, cout), nn.Sigmoid())

    def forward(self, x):
        t = self.conv1(x)
        for r in range(2):
            t = torch.cat([self.mid1_conv[r](t), self.mid2_conv[r](t)], dim=1)
        y = self.conv2(t + x)
        return y * self.att(y)[:, :, None, None]


class _HourglassM(nn.Module):
    """:93-138 (same data flow as variant A's hourglass)."""

    def __init__(self, levels, c, blocks, p_drop):
        super().__init__()
        assert len(blocks) == levels - 1
        self.num_levels = levels
        self.encoder = nn.ModuleList([ME_att(c, c, p_drop)])
        self.decoder = nn.ModuleList()
        for nb in blocks:
            self.encoder.append(MyResidual(c, c, 2, nb))
            self.decoder.append(MyResidual(c, c))
        self.decoder.append(ME_att(c, c, p_drop))

    forward = _HourglassA.forward


class _StemM(nn.Module):
    """my_pelee_stem :196-228."""

    def __init__(self, cout, min_mid=32):
        super().__init__()
        m = max(cout // 4, min_mid)
        self.conv1 = _seq(*_cbr(3, m, 3, 2, 1, bias=False, act=nn.LeakyReLU()),
                          *_cbr(m, m, 3, 1, 1, m, False, act=nn.LeakyReLU()))
        self.branch1 = _seq(*_cbr(m, m, 1, act=nn.ReLU()), *_cbr(m, m, 3, 2, 1, act=nn.ReLU()))
        self.branch2 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.conv1x1 = nn.Conv2d(2 * m, cout, 1)

    forward = _StemA.forward


class MultiScaleAttentionHourglass(nn.Module):
    """:231-265.  cfg keys: MODEL.{num_stage,input_channel,output_channel,num_block,output_acitivation}."""

    def __init__(self, cfg, p_drop=0.3):
        super().__init__()
        M = cfg.MODEL
        c = M.get("input_channel", 128)
        self.with_activation = M.get("output_acitivation", False)
        self.pre = _StemM(c)
        self.hgs = _HourglassM(M.get("num_stage", 4), c, M.get("num_block", [2, 2, 2]), p_drop)
        self.features = _seq(MyBottleNeck(c), *_cbr(c, c, 1, act=nn.LeakyReLU()))
        self.outs = nn.Conv2d(c, M.get("output_channel", cfg.DATASET.num_joints), 1)
        self.init_weights()

    def forward(self, x):
        y = self.outs(self.features(self.hgs(self.pre(x))[-1]))
        return F.leaky_relu(y, 0.5) if self.with_activation else y

    def init_weights(self):  # :256-262: conv weight ~ N(0,1), bias 0 (weight_init.py:28-32); BN gamma 1, beta 0
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 1)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)


# --------------------------------------------------------------------------
# hourglass baseline (models/pose_estimation/hourglassnet.py) -- BASELINE config 5
# --------------------------------------------------------------------------
class HGConv(nn.Module):
    """hourglassnet.py:6-25: biased conv (pad (k-1)//2) [+ BatchNorm] [+ ReLU]."""

    def __init__(self, inp_dim, out_dim, kernel_size=3, stride=1, bn=False, relu=True):
        super().__init__()
        self.inp_dim = inp_dim
        self.conv = nn.Conv2d(inp_dim, out_dim, kernel_size, stride, padding=(kernel_size - 1) // 2, bias=True)
        self.relu = nn.ReLU() if relu else None
        self.bn = nn.BatchNorm2d(out_dim) if bn else None

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return self.relu(x) if self.relu is not None else x


class HGResidual(nn.Module):
    """hourglassnet.py:27-53: pre-activation bottleneck BN-ReLU-1x1, BN-ReLU-3x3, BN-ReLU-1x1, plus (projected) skip."""

    def __init__(self, inp_dim, out_dim):
        super().__init__()
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm2d(inp_dim)
        self.conv1 = HGConv(inp_dim, out_dim // 2, 1, relu=False)
        self.bn2 = nn.BatchNorm2d(out_dim // 2)
        self.conv2 = HGConv(out_dim // 2, out_dim // 2, 3, relu=False)
        self.bn3 = nn.BatchNorm2d(out_dim // 2)
        self.conv3 = HGConv(out_dim // 2, out_dim, 1, relu=False)
        self.skip_layer = nn.Identity() if inp_dim == out_dim else HGConv(inp_dim, out_dim, 1, relu=False)

    def forward(self, x):
        out = self.conv1(self.relu(self.bn1(x)))
        out = self.conv2(self.relu(self.bn2(out)))
        out = self.conv3(self.relu(self.bn3(out)))
        return out + self.skip_layer(x)


class HourglassModule(nn.Module):
    """hourglassnet.py:55-81 (recursive)."""

    def __init__(self, n, f, bn=None, increase=0):
        super().__init__()
        nf = f + increase
        self.up1 = HGResidual(f, f)
        self.pool1 = nn.MaxPool2d(2, 2)
        self.low1 = HGResidual(f, nf)
        self.n = n
        self.low2 = HourglassModule(n - 1, nf, bn=bn) if n > 1 else HGResidual(nf, nf)
        self.low3 = HGResidual(nf, f)
        self.up2 = nn.Upsample(scale_factor=2, mode="nearest")

    def forward(self, x):
        return self.up1(x) + self.up2(self.low3(self.low2(self.low1(self.pool1(x)))))


class HGMerge(nn.Module):
    def __init__(self, x_dim, y_dim):
        super().__init__()
        self.conv = HGConv(x_dim, y_dim, 1, relu=False, bn=False)

    def forward(self, x):
        return self.conv(x)


class HourglassNet(nn.Module):
    """hourglassnet.py:90-136: stacked hourglass, output [N, num_stack, K, H/4, W/4]."""

    def __init__(self, cfg):
        super().__init__()
        M = cfg.MODEL
        ns, nl = M.get("num_stack", 8), M.get("num_level", 4)
        c, k = M.get("input_channel", 256), M.get("output_channel", 21)
        self.num_stack = ns
        self.pre = nn.Sequential(HGConv(3, 64, 7, 2, bn=True, relu=True), HGResidual(64, 128), nn.MaxPool2d(2, 2),
                                 HGResidual(128, 128), HGResidual(128, c))
        self.hgs = nn.ModuleList([nn.Sequential(HourglassModule(nl, c, bn=False, increase=0)) for _ in range(ns)])
        self.features = nn.ModuleList([nn.Sequential(HGResidual(c, c), HGConv(c, c, 1, bn=True, relu=True)) for _ in range(ns)])
        self.outs = nn.ModuleList([HGConv(c, k, 1, relu=False, bn=False) for _ in range(ns)])
        self.merge_features = nn.ModuleList([HGMerge(c, c) for _ in range(ns - 1)])
        self.merge_preds = nn.ModuleList([HGMerge(k, c) for _ in range(ns - 1)])

    def forward(self, imgs):
        x = self.pre(imgs)
        outs = []
        for i in range(self.num_stack):
            feature = self.features[i](self.hgs[i](x))
            preds = self.outs[i](feature)
            outs.append(preds)
            if i < self.num_stack - 1:
                x = x + self.merge_preds[i](preds) + self.merge_features[i](feature)
        return torch.stack(outs, dim=1)


# --------------------------------------------------------------------------
# Lite-HRNet baseline (models/pose_estimation/lite_hrnet.py) -- BASELINE config 5
# --------------------------------------------------------------------------
class LHDWConv(nn.Module):
    """lite_hrnet.py:11-27: depthwise 3x3 + BN [+ ReLU] -> 1x1 + BN [+ ReLU]."""

    def __init__(self, cin, cout, stride=1, padding=1, dilation=1, mid_relu=True, last_relu=True, bias=False):
        super().__init__()
        self.depthwise_conv = nn.Sequential(nn.Conv2d(cin, cin, 3, stride, padding, groups=cin, bias=bias, dilation=dilation),
                                            nn.BatchNorm2d(cin))
        self.mid_relu = nn.ReLU() if mid_relu else nn.Identity()
        self.pointwise_conv = nn.Sequential(nn.Conv2d(cin, cout, 1, 1, 0, bias=bias), nn.BatchNorm2d(cout))
        self.last_relu = nn.ReLU() if last_relu else nn.Identity()

    def forward(self, x):
        return self.last_relu(self.pointwise_conv(self.mid_relu(self.depthwise_conv(x))))


def channel_shuffle(x, groups):
    """lite_hrnet.py:29-52."""
    b, c, h, w = x.size()
    return x.view(b, groups, c // groups, h, w).transpose(1, 2).contiguous().view(b, -1, h, w)


class SpatialWeighting(nn.Module):
    """lite_hrnet.py:55-74: x * sigmoid(relu(conv2(sigmoid(relu(conv1(global_avg_pool(x)))))))."""

    def __init__(self, channels, ratio=16):
        super().__init__()
        self.global_avgpool = nn.AdaptiveAvgPool2d(1)
        mid = int(channels / ratio)
        self.conv1 = nn.Sequential(nn.Conv2d(channels, mid, 1, 1), nn.ReLU(True), nn.Sigmoid())
        self.conv2 = nn.Sequential(nn.Conv2d(mid, channels, 1, 1), nn.ReLU(True), nn.Sigmoid())

    def forward(self, x):
        return x * self.conv2(self.conv1(self.global_avgpool(x)))


class CrossResolutionWeighting(nn.Module):
    """lite_hrnet.py:76-108."""

    def __init__(self, channels, ratio=16):
        super().__init__()
        self.channels = channels
        total = sum(channels)
        mid = int(total / ratio)
        self.conv1 = nn.Sequential(nn.Conv2d(total, mid, 1, 1), nn.BatchNorm2d(mid), nn.ReLU(True), nn.Sigmoid())
        self.conv2 = nn.Sequential(nn.Conv2d(mid, total, 1, 1), nn.BatchNorm2d(total), nn.ReLU(True), nn.Sigmoid())

    def forward(self, x):
        mini = x[-1].shape[-2:]
        out = torch.cat([F.adaptive_avg_pool2d(s, mini) for s in x[:-1]] + [x[-1]], dim=1)
        out = torch.split(self.conv2(self.conv1(out)), self.channels, dim=1)
        return [s * F.interpolate(a, size=s.shape[-2:], mode="nearest") for s, a in zip(x, out)]


class ConditionalChannelWeighting(nn.Module):
    """lite_hrnet.py:110-143."""

    def __init__(self, in_channels, reduce_ratio, stride=1):
        super().__init__()
        bc = [c // 2 for c in in_channels]
        self.cross_resolution_weighting = CrossResolutionWeighting(channels=bc, ratio=reduce_ratio)
        self.depthwise_convs = nn.ModuleList([nn.Sequential(nn.Conv2d(c, c, 3, stride, 1, groups=c), nn.BatchNorm2d(c)) for c in bc])
        self.spatial_weighting = nn.ModuleList([SpatialWeighting(channels=c, ratio=4) for c in bc])

    def forward(self, x):
        x = [s.chunk(2, dim=1) for s in x]
        x1, x2 = [s[0] for s in x], [s[1] for s in x]
        x2 = self.cross_resolution_weighting(x2)
        x2 = [dw(s) for s, dw in zip(x2, self.depthwise_convs)]
        x2 = [sw(s) for s, sw in zip(x2, self.spatial_weighting)]
        return [channel_shuffle(torch.cat([a, b], dim=1), 2) for a, b in zip(x1, x2)]


class StageModule(nn.Module):
    """lite_hrnet.py:145-204."""

    def __init__(self, in_branches, num_blocks, in_channels, reduce_ratio=8, with_fuse=True):
        super().__init__()
        self.in_branches, self.in_channels, self.with_fuse = in_branches, in_channels, with_fuse
        self.layers = nn.Sequential(*[ConditionalChannelWeighting(in_channels, reduce_ratio) for _ in range(num_blocks)])
        if self.with_fuse and self.in_branches > 1:
            self.fuse_layers = self._make_fuse_layers()
            self.relu = nn.ReLU()
        else:
            self.with_fuse = False

    def _make_fuse_layers(self):
        cs = self.in_channels
        fuse = nn.ModuleList()
        for i in range(self.in_branches):
            fuse.append(nn.ModuleList())
            for j in range(self.in_branches):
                cin, cout = cs[j], cs[i]
                if i == j:
                    fuse[-1].append(nn.Identity())
                elif j > i:
                    fuse[-1].append(nn.Sequential(nn.Conv2d(cin, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout),
                                                  nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                else:
                    down = [LHDWConv(cin, cin, stride=2, mid_relu=False, last_relu=False) for _ in range(i - j - 1)]
                    down.append(LHDWConv(cin, cout, stride=2, mid_relu=False, last_relu=False))
                    fuse[-1].append(nn.Sequential(*down))
        return fuse

    def forward(self, x):
        if self.in_branches == 1:
            return [self.layers[0](x[0])]
        out = self.layers(x)
        if self.with_fuse:
            # lite_hrnet.py:190-199, including its aliasing: for i == 0 `y` IS out[0] and `y += ...` accumulates INTO it, so
            #   out[0] becomes 2*out[0] + sum_j fuse[0][j](out[j])   (the un-rectified fused branch 0),
            # and every later row i starts from fuse[i][0](that), evaluated TWICE (initial value + the j == 0 term; in
            # training mode its BatchNorm running statistics therefore move twice per forward).
            out = list(out)
            out_fuse = []
            for i in range(len(self.fuse_layers)):
                y = out[0] if i == 0 else self.fuse_layers[i][0](out[0])
                for j in range(self.in_branches):
                    y = y + (out[j] if i == j else self.fuse_layers[i][j](out[j]))
                if i == 0:
                    out[0] = y
                out_fuse.append(self.relu(y))
            out = out_fuse
        return out


class LHStemModule(nn.Module):
    """lite_hrnet.py:206-248."""

    def __init__(self, in_channels, stem_channels, out_channels, expand_ratio):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, stem_channels, 3, 2, 1), nn.BatchNorm2d(stem_channels), nn.ReLU())
        self.out_channels = out_channels
        mid = int(round(stem_channels * expand_ratio))
        bc = stem_channels // 2
        inc = out_channels - bc if stem_channels == out_channels else out_channels - stem_channels
        self.branch1 = LHDWConv(bc, inc, stride=2, mid_relu=False, bias=True)
        self.expand_conv = nn.Sequential(nn.Conv2d(bc, mid, 1, 1, 0), nn.BatchNorm2d(mid), nn.ReLU())
        self.depthwise_conv = nn.Sequential(nn.Conv2d(mid, mid, 3, 2, 1, groups=mid), nn.BatchNorm2d(mid))
        lc = bc if stem_channels == out_channels else stem_channels
        self.linear_conv = nn.Sequential(nn.Conv2d(mid, lc, 1, 1, 0), nn.BatchNorm2d(lc), nn.ReLU())

    def forward(self, x):
        x = self.conv1(x)
        x1, x2 = x.chunk(2, dim=1)
        x2 = self.linear_conv(self.depthwise_conv(self.expand_conv(x2)))
        return channel_shuffle(torch.cat((self.branch1(x1), x2), dim=1), 2)


class IterativeHead(nn.Module):
    """lite_hrnet.py:250-281."""

    def __init__(self, in_channels):
        super().__init__()
        nb = len(in_channels)
        self.in_channels = in_channels[::-1]
        self.projects = nn.ModuleList([LHDWConv(self.in_channels[i], self.in_channels[i + 1] if i != nb - 1 else self.in_channels[i])
                                       for i in range(nb)])

    def forward(self, x):
        x = x[::-1]
        y, last = [], None
        for i, s in enumerate(x):
            if last is not None:
                s = s + F.interpolate(last, size=s.shape[-2:], mode="bilinear", align_corners=True)
            s = self.projects[i](s)
            y.append(s)
            last = s
        return y[::-1]


class LiteHRNet(nn.Module):
    """lite_hrnet.py:284-390."""

    def __init__(self, cfg):
        super().__init__()
        out_channel = cfg.MODEL.get("output_channel", cfg.DATASET.num_joints)
        depth = cfg.MODEL.get("depth", 30)
        self.stem = LHStemModule(in_channels=3, stem_channels=32, out_channels=32, expand_ratio=1)
        self.num_stages = 3
        self.stages_spec = dict(num_modules=(3, 8, 3) if depth != 18 else (3, 4, 3), num_branches=(2, 3, 4), num_blocks=(2, 2, 2),
                                with_fuse=(True, True, True), reduce_ratios=(8, 8, 8),
                                num_channels=((40, 80), (40, 80, 160), (40, 80, 160, 320)))
        last = [self.stem.out_channels]
        for i in range(self.num_stages):
            nc = list(self.stages_spec["num_channels"][i])
            setattr(self, f"transition{i}", self._make_transition_layer(last, nc))
            stage, last = self._make_stage(i, nc)
            setattr(self, f"stage{i}", stage)
        self.head_layer = IterativeHead(in_channels=last)
        self.out_conv = nn.Conv2d(40, out_channel, 1, 1, 0)

    @staticmethod
    def _make_transition_layer(pre, cur):
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                layers.append(LHDWConv(pre[i], cur[i], mid_relu=False) if cur[i] != pre[i] else None)
            else:
                down = []
                for j in range(i + 1 - len(pre)):
                    cin = pre[-1]
                    down.append(LHDWConv(cin, cur[i] if j == i - len(pre) else cin, stride=2, mid_relu=False))
                layers.append(nn.Sequential(*down))
        return nn.ModuleList(layers)

    def _make_stage(self, si, in_channels):
        sp = self.stages_spec
        mods = []
        for _ in range(sp["num_modules"][si]):
            mods.append(StageModule(sp["num_branches"][si], sp["num_blocks"][si], in_channels, sp["reduce_ratios"][si], sp["with_fuse"][si]))
            in_channels = mods[-1].in_channels
        return nn.Sequential(*mods), in_channels

    def forward(self, x):
        y_list = [self.stem(x)]
        for i in range(self.num_stages):
            tr = getattr(self, f"transition{i}")
            x_list = []
            for j in range(self.stages_spec["num_branches"][i]):
                if tr[j]:
                    x_list.append(tr[j](y_list[-1] if j >= len(y_list) else y_list[j]))
                else:
                    x_list.append(y_list[j])
            y_list = getattr(self, f"stage{i}")(x_list)
        return self.out_conv(self.head_layer(y_list)[0])


def get_model(cfg, p_drop=0.3):
    """Mirror of models/__init__.py:20-26 restricted to the hot path.

    `litehandnet` -> variant A (the registered class); `litehourglass` -> variant B
    (unregistered in the reference; file litehourglass.py)."""
    name = cfg.MODEL.name
    if name == "litehandnet":
        return LiteHandNet(cfg, p_drop)
    if name == "litehourglass":
        return LiteHourglassNet(cfg, p_drop)
    if name == "mynet":
        return MultiScaleAttentionHourglass(cfg, p_drop)
    if name == "hourglass":
        return HourglassNet(cfg)
    if name == "litehrnet":
        return LiteHRNet(cfg)
    raise AssertionError(f"model <{name}> is outside the hot path")


# --------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------
def distance_loss(output, target, target_weight, balance=True, thr=0.5):
    """Class-balanced weighted MSE, loss/heatmapLoss.py:242-265 (L2, reduction='mean')."""
    l = (output - target) ** 2 * target_weight.unsqueeze(-1)
    if balance:
        pos = target > thr
        n = l.numel()
        pf = n / (pos.sum() + 1) * 0.1
        nf = n / ((~pos).sum() + 1)
        l = torch.where(pos, l * pf, l * nf)
    return l.mean()


class SimDRLoss(nn.Module):
    """loss/centernet_simdr_loss.py:6-71.  Per joint: SmoothL1 (mean over [N, L]) times the MEAN of that joint's weights
    (the reference multiplies a scalar by the weight vector and takes the mean), x and y, averaged over the joints."""

    def __init__(self, cfg):
        super().__init__()
        k = cfg.PIPELINE.simdr_split_ratio
        self.simdr_width, self.simdr_height = int(k * cfg.DATASET.image_size[0]), int(k * cfg.DATASET.image_size[1])
        feat = int(cfg.DATASET.heatmap_size[0] * cfg.DATASET.heatmap_size[1])
        self.x_shared_decoder = nn.Linear(feat, self.simdr_width)
        self.y_shared_decoder = nn.Linear(feat, self.simdr_height)

    def forward(self, heatmap, simdr_x, simdr_y, target_weight):
        flat = heatmap.flatten(start_dim=2)
        px, py = self.x_shared_decoder(flat), self.y_shared_decoder(flat)
        K = px.size(1)
        total = 0
        for j in range(K):
            mw = target_weight[:, j].reshape(-1).mean()
            total = total + F.smooth_l1_loss(px[:, j], simdr_x[:, j]) * mw + F.smooth_l1_loss(py[:, j], simdr_y[:, j]) * mw
        return total / K


class TopdownHeatmapLoss(nn.Module):
    """loss/loss.py:69-114 with auto_weight False (SimDR auxiliary loss when simdr_split_ratio > 0)."""

    def __init__(self, cfg):
        super().__init__()
        self.balance = cfg.MODEL.name != "atthandnet"
        self.loss_weight = cfg.LOSS.loss_weight
        self.simdr_loss = SimDRLoss(cfg) if cfg.PIPELINE.simdr_split_ratio > 0 else None

    def forward(self, output, meta):
        t = meta["target"].to(output.device)
        w = meta["target_weight"].to(output.device)
        loss = self.loss_weight[0] * distance_loss(output, t, w, self.balance)
        d = {"heatmap": loss.item()}
        if self.simdr_loss is not None:
            ls = self.loss_weight[1] * self.simdr_loss(output, meta["simdr_x"], meta["simdr_y"], w)
            d["simdr"] = ls.item()
            loss = loss + ls
        return loss, d

"""HIP-backed mirror of models/pose_estimation/liteHandNet/liteHandNet.py -- the registered `litehandnet`
("variant A": Stem -> MSAB / Residual hourglass -> BottleNeck neck -> 1x1 head).  Same constructor arguments,
attribute names and state_dict keys as the reference."""
from torch import nn

from .common import ChannelAttension, SEBlock
from .engine import PlanModule
from .repblocks import RepBlock, RepConv, act_slope


class DWConv(PlanModule):
    """liteHandNet.py:8-21: depthwise 3x3 RepConv -> pointwise 1x1 RepConv."""

    def __init__(self, in_channel, out_channel, stride=1, padding=1, dilation=1, activation=nn.LeakyReLU):
        super().__init__()
        self.depthwise_conv = RepConv(in_channel, in_channel, 3, stride, padding, groups=in_channel, dilation=dilation,
                                      activation=activation, inplace=False)
        self.pointwise_conv = RepConv(in_channel, out_channel, 1, 1, 0, activation=activation, inplace=False)

    def emit(self, pb, x, out=None):
        return self.pointwise_conv.emit(pb, self.depthwise_conv.emit(pb, x), out=out)


class BottleNeck(PlanModule):
    """liteHandNet.py:23-37: act(x + 1x1 -> 3x3 -> 1x1)."""

    def __init__(self, channel, reduction=4, activation=nn.LeakyReLU):
        super().__init__()
        m = channel // reduction
        self.conv = nn.Sequential(RepConv(channel, m, 1, 1, 0, activation=activation, inplace=True),
                                  RepConv(m, m, 3, 1, 1, activation=activation, inplace=True),
                                  RepConv(m, channel, 1, 1, 0, activation=None))
        self.activation = activation()
        self.out_slope = act_slope(activation, positional=False)

    def emit(self, pb, x, out=None):
        t = x
        for c in self.conv:
            t = c.emit(pb, t)
        return pb.ew([x, t], out_slope=self.out_slope, out=out)


class BasicBlock(PlanModule):
    """liteHandNet.py:39-54: act(skip(x) + 3x3(s) -> 3x3)."""

    def __init__(self, inp_dim, out_dim, stride=1, activation=nn.LeakyReLU):
        super().__init__()
        self.conv = nn.Sequential(RepConv(inp_dim, out_dim, 3, stride, 1, activation=activation, inplace=True),
                                  RepConv(inp_dim, out_dim, 3, 1, 1, activation=None))
        if stride == 2 or inp_dim != out_dim:
            self.skip_layer = RepConv(inp_dim, out_dim, 1, stride, 0, activation=None)
        else:
            self.skip_layer = nn.Identity()
        self.activation = activation()
        self.out_slope = act_slope(activation, positional=False)

    def emit(self, pb, x, out=None):
        t = self.conv[1].emit(pb, self.conv[0].emit(pb, x))
        s = self.skip_layer.emit(pb, x) if isinstance(self.skip_layer, RepConv) else x
        return pb.ew([s, t], out_slope=self.out_slope, out=out)


class Residual(PlanModule):
    """liteHandNet.py:57-68."""

    def __init__(self, inp_dim, out_dim, stride=2, num_block=2, reduction=2, activation=nn.LeakyReLU):
        super().__init__()
        self.conv1 = BasicBlock(inp_dim, out_dim, stride, activation)
        self.blocks = nn.Sequential(*[BottleNeck(out_dim, reduction, activation) for _ in range(num_block)])

    def emit(self, pb, x, out=None):
        x = self.conv1.emit(pb, x)
        for b in self.blocks:
            x = b.emit(pb, x)
        return x


class MSAB(PlanModule):
    """liteHandNet.py:116-166."""

    def __init__(self, in_c, out_c, ca_type="ca", activation=nn.LeakyReLU, p_drop=0.3):
        super().__init__()
        m = in_c // 2
        a = activation
        self.conv1 = RepConv(in_c, m, 1, 1, 0, activation=a, inplace=True)
        self.mid1_conv = nn.ModuleList([
            nn.Sequential(DWConv(m, m // 2, activation=a), DWConv(m // 2, m // 2, activation=a)),
            nn.Sequential(DWConv(m, m, activation=a), DWConv(m, m, activation=a))])
        self.mid2_conv = nn.ModuleList([
            nn.Sequential(DWConv(m, m // 2, dilation=2, padding=2, activation=a), DWConv(m // 2, m // 2, activation=a)),
            nn.Sequential(DWConv(m, m, dilation=2, padding=2, activation=a), DWConv(m, m, activation=a))])
        self.conv2 = RepConv(in_c, out_c, 1, 1, 0, activation=a, inplace=True)
        if ca_type == "ca":
            self.ca = ChannelAttension(out_c, p_drop=p_drop)
        elif ca_type == "none":
            self.ca = nn.Identity()
        elif ca_type == "se":
            self.ca = SEBlock(out_c, internal_neurons=out_c // 16)      # liteHandNet.py:147-148
        else:
            raise ValueError(f"<ca_type={ca_type!r}> not in se|ca|none")
        self.mid_c = m

    def emit(self, pb, x, out=None):
        m = self.conv1.emit(pb, x)
        for r in range(2):
            half = self.mid_c // 2 if r == 0 else self.mid_c
            cat = pb.new(m.H, m.W, 2 * half)
            for j, branch in enumerate((self.mid1_conv[r], self.mid2_conv[r])):
                t = branch[0].emit(pb, m)
                branch[1].emit(pb, t, out=pb.slice(cat, j * half, half))
            m = cat
        y = self.conv2.emit(pb, pb.ew([m, x], lazy=True))      # `m + x` is summed on load by the 1x1 (never written in forward)
        if isinstance(self.ca, (ChannelAttension, SEBlock)):
            y = self.ca.emit(pb, y)
        return y


class EncoderDecoder(PlanModule):
    """liteHandNet.py:71-113."""

    def __init__(self, num_levels=5, inp_dim=128, num_blocks=[], ca_type="ca", reduction=2, activation=nn.LeakyReLU,
                 p_drop=0.3):
        super().__init__()
        assert len(num_blocks) == num_levels - 1
        self.num_levels = num_levels
        self.encoder, self.decoder = nn.ModuleList([]), nn.ModuleList([])
        self.encoder.append(MSAB(inp_dim, inp_dim, ca_type=ca_type, p_drop=p_drop))
        for i in range(num_levels - 1):
            self.encoder.append(Residual(inp_dim, inp_dim, 2, num_blocks[i], reduction, activation))
            self.decoder.append(Residual(inp_dim, inp_dim, 1, num_blocks[i], reduction, activation))
        self.decoder.append(MSAB(inp_dim, inp_dim, ca_type=ca_type, p_drop=p_drop))

    def emit(self, pb, x, out=None):
        enc = []
        for layer in self.encoder:
            x = layer.emit(pb, x)
            enc.append(x)
        short = pb.avgpool(enc[0], enc[-1].H, enc[-1].W)
        for i, layer in enumerate(self.decoder):
            peer = enc[self.num_levels - 1 - i]
            if i == 0:
                x = pb.ew([layer.emit(pb, peer), short])
            else:
                x = pb.ew([layer.emit(pb, x), peer])        # nearest upsample + add
        return x


class Stem(PlanModule):
    """liteHandNet.py:169-193."""
    consumes_image = True

    def __init__(self, out_channel=256, min_mid_c=32, activation=nn.LeakyReLU):
        super().__init__()
        m = out_channel // 4 if out_channel // 4 >= min_mid_c else min_mid_c
        self.conv1 = nn.Sequential(RepBlock(3, m, 3, 2, 1, activation=activation, inplace=True),
                                   RepBlock(m, m, 7, 1, 3, groups=m, activation=activation, inplace=True))
        self.branch1 = nn.Sequential(RepConv(m, m, 1, 1, 0, activation=activation, inplace=True),
                                     RepConv(m, m, 3, 2, 1, activation=activation, inplace=True))
        self.branch2 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.conv1x1 = nn.Conv2d(m * 2, out_channel, 1, 1, 0)
        self.mid = m

    def emit(self, pb, x, out=None):
        m = self.mid
        t = self.conv1[1].emit(pb, self.conv1[0].emit(pb, x))
        cat = pb.new((t.H + 1) // 2, (t.W + 1) // 2, 2 * m)
        self.branch1[1].emit(pb, self.branch1[0].emit(pb, t), out=pb.slice(cat, 0, m))
        pb.maxpool(t, out=pb.slice(cat, m, m))
        return pb.conv(cat, self.conv1x1, None, out=out)


class LiteHandNet(PlanModule):
    """liteHandNet.py:196-244.  cfg.MODEL keys: num_stage, input_channel, output_channel, num_block, ca_type,
    reduction, activation."""
    consumes_image = True

    def __init__(self, cfg):
        super().__init__()
        M = cfg.MODEL
        num_stage = M.get("num_stage", 4)
        inp_dim = M.get("input_channel", 128)
        oup_dim = M.get("output_channel", cfg.DATASET.num_joints)
        num_block = M.get("num_block", [2, 2, 2])
        ca_type = M.get("ca_type", "ca")
        reduction = M.get("reduction", 2)
        activation = M.get("activation", "LeakyReLU")
        assert reduction in [2, 4]
        assert ca_type in ["ca", "se", "none"]
        assert activation.lower() in ["leakyrelu", "relu", "silu"]
        activation = {"leakyrelu": nn.LeakyReLU, "relu": nn.ReLU, "silu": nn.SiLU}[activation.lower()]
        self.p_drop = float(M.get("ca_dropout", 0.3))
        self.pre = Stem(inp_dim, activation=activation)
        self.hgs = EncoderDecoder(num_stage, inp_dim, num_block, ca_type, reduction, activation, p_drop=self.p_drop)
        self.features = nn.Sequential(BottleNeck(inp_dim, 2, activation),
                                      RepConv(inp_dim, inp_dim, 1, 1, 0, activation=activation, inplace=True))
        self.out_layer = nn.Conv2d(inp_dim, oup_dim, 1, 1, 0)
        self.init_weights()

    def emit(self, pb, x, out=None):
        y = self.hgs.emit(pb, self.pre.emit(pb, x))
        for m in self.features:
            y = m.emit(pb, y)
        return pb.conv(y, self.out_layer, None, nchw_out=True)

    def init_weights(self):
        # liteHandNet.py:236-238 + weight_init.py:28-32: EVERY module with a .weight (conv AND BatchNorm) ~ N(0,1), bias 0
        for m in self.modules():
            w = getattr(m, "weight", None)
            if w is not None and hasattr(w, "data"):
                nn.init.normal_(w, 0, 1)
            b = getattr(m, "bias", None)
            if b is not None and hasattr(b, "data"):
                nn.init.constant_(b, 0)

    def deploy_model(self):
        """Re-parameterise every unit that can (reference deploy_model; called by test.py:106-107)."""
        for m in self.modules():
            if hasattr(m, "switch_to_deploy"):
                m.switch_to_deploy()
        self.deploy = True
        self.__dict__.pop("_engine", None)

"""HIP-backed mirror of models/pose_estimation/liteHandNet/litehourglass.py -- the MSRB hourglass
("variant B").  Same attribute names / state_dict keys as the reference."""
from torch import nn

from . import _lib
from .common import ChannelAttension, SEBlock
from .engine import PlanModule
from .repblocks import RepConv


def _make_ca(ca_type, channels, p_drop):
    if ca_type == "ca":
        return ChannelAttension(channels, p_drop=p_drop)
    if ca_type == "se":
        return SEBlock(channels, internal_neurons=channels // 16)       # litehourglass.py:33-35, 64-66
    return nn.Identity()


class MSRB(PlanModule):
    """litehourglass.py:13-50: 2 x [chunk -> dw3x3(d1) || dw3x3(d2) -> cat -> CA -> add] -> 1x1."""

    def __init__(self, in_channels, out_channels, ca_type="none", p_drop=0.3):
        super().__init__()
        h = self.half_channels = in_channels // 2
        self.branch1 = nn.ModuleList([RepConv(h, h, 3, 1, 1, groups=h, activation=None) for _ in range(2)])
        self.branch2 = nn.ModuleList([RepConv(h, h, 3, 1, 2, 2, groups=h, activation=None) for _ in range(2)])
        self.ca = nn.ModuleList([_make_ca(ca_type, out_channels, p_drop) for _ in range(2)])
        self.conv = RepConv(in_channels, out_channels, 1, 1, 0)

    def emit(self, pb, x, out=None):
        h = self.half_channels
        acc = x
        for r, (b1, b2, ca) in enumerate(zip(self.branch1, self.branch2, self.ca)):
            lr = pb.new(acc.H, acc.W, 2 * h)
            b1.emit(pb, pb.slice(acc, 0, h), out=pb.slice(lr, 0, h))
            b2.emit(pb, pb.slice(acc, h, h), out=pb.slice(lr, h, h))
            if isinstance(ca, (ChannelAttension, SEBlock)):
                lr = ca.emit(pb, lr)
            # `out + ca(cat)` (and `out + x` on the last round) are never written in forward: the next round's depthwise
            # kernels and the closing 1x1 add the operands while loading them
            acc = pb.ew([acc, lr] + ([x] if r == 1 else []), lazy=True)
        return self.conv.emit(pb, acc, out=out)


class RepBasicUnit(PlanModule):
    """litehourglass.py:52-78: keep the left half, right half -> 1x1 -> dw3x3, cat, CA/Identity."""

    def __init__(self, in_channels, out_channels, ca_type="ca", p_drop=0.3):
        super().__init__()
        self.left_part = in_channels // 2
        self.right_part_in = in_channels - self.left_part
        self.right_part_out = out_channels - self.left_part
        self.conv = nn.Sequential(
            RepConv(self.right_part_in, self.right_part_out, kernel=1),
            RepConv(self.right_part_out, self.right_part_out, kernel=3, padding=1, groups=self.right_part_out))
        if ca_type not in ("ca", "se", "none"):
            raise ValueError(f"<ca_type={ca_type!r}> not in se|ca|none")
        self.ca = _make_ca(ca_type, out_channels, p_drop)

    def emit(self, pb, x, out=None):
        L = self.left_part
        gated = isinstance(self.ca, (ChannelAttension, SEBlock))
        if not gated and out is None:
            # cat(left, branch(right)) as a two-part tensor: the pass-through half is never copied (its consumers read it
            # where it lives, its gradient lands in the producer's gradient buffer directly)
            t = self.conv[0].emit(pb, pb.slice(x, L, self.right_part_in))
            return pb.cat([pb.slice(x, 0, L), self.conv[1].emit(pb, t)])
        y = out if (out is not None and not gated) else pb.new(x.H, x.W, L + self.right_part_out)
        pb.ew([pb.slice(x, 0, L)], out=pb.slice(y, 0, L))                       # left half passes through
        t = self.conv[0].emit(pb, pb.slice(x, L, self.right_part_in))
        self.conv[1].emit(pb, t, out=pb.slice(y, L, self.right_part_out))
        if gated:
            y = self.ca.emit(pb, y)
            if out is not None:
                y = pb.ew([y], out=out)
        return y


class EncoderDecoder(PlanModule):
    """litehourglass.py:108-163."""

    def __init__(self, num_stage=4, channel=128, msrb_ca="ca", rbu_ca="ca", p_drop=0.3):
        super().__init__()
        self.num_stage = num_stage
        self.encoder, self.decoder = nn.ModuleList([]), nn.ModuleList([])
        self.maxpool = nn.MaxPool2d(2, 2)
        for i in range(num_stage):
            for lst in (self.encoder, self.decoder):
                first = (MSRB(channel, channel, ca_type=msrb_ca, p_drop=p_drop) if i == 0
                         else RepBasicUnit(channel, channel, ca_type=rbu_ca, p_drop=p_drop))
                lst.append(nn.Sequential(first, RepBasicUnit(channel, channel, ca_type=rbu_ca, p_drop=p_drop)))

    def emit(self, pb, x, out=None):
        skips = []
        for i in range(self.num_stage):
            for m in self.encoder[i]:
                x = m.emit(pb, x)
            skips.append(x)
            if i != self.num_stage - 1:
                x = pb.maxpool(x)
        for i in range(self.num_stage - 1, -1, -1):
            if i == self.num_stage - 1:
                x = skips[i]
                for m in self.decoder[i]:
                    x = m.emit(pb, x)
                x = pb.ew([x, pb.avgpool(skips[0], x.H, x.W)])
            else:
                x = pb.ew([x, skips[i]])                 # nearest upsample + add in one pass
                for m in self.decoder[i]:
                    x = m.emit(pb, x)
        return x


class Stem(PlanModule):
    """litehourglass.py:166-193."""
    consumes_image = True

    def __init__(self, channel, p_drop=0.3):
        super().__init__()
        m = max(channel // 4, 32)
        self.conv1 = nn.Sequential(RepConv(3, m, 3, 2, 1), RepConv(m, m, 3, 1, 1, groups=m))
        self.branch1 = nn.Sequential(RepConv(m, m, 1, 1, 0), RepConv(m, m, 3, 2, 1, groups=m, activation=None),
                                     RepConv(m, m, 1, 1, 0))
        self.branch2 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.conv2 = nn.Sequential(RepConv(2 * m, channel), RepBasicUnit(channel, channel, p_drop=p_drop),
                                   RepBasicUnit(channel, channel, p_drop=p_drop))
        self.mid = m

    def emit(self, pb, x, out=None):
        m = self.mid
        t = self.conv1[1].emit(pb, self.conv1[0].emit(pb, x))
        cat = pb.new((t.H + 1) // 2, (t.W + 1) // 2, 2 * m)
        b = self.branch1[1].emit(pb, self.branch1[0].emit(pb, t))
        self.branch1[2].emit(pb, b, out=pb.slice(cat, 0, m))
        pb.maxpool(t, out=pb.slice(cat, m, m))
        y = cat
        for mod in self.conv2:
            y = mod.emit(pb, y)
        return y


class LiteHandNet(PlanModule):
    """litehourglass.py:196-237.  cfg.MODEL keys: num_stage, msrb_ca, rbu_ca, input_channel, output_channel."""
    consumes_image = True

    def __init__(self, cfg, deploy=False):
        super().__init__()
        if deploy:
            raise _lib.LhnError("deploy form is not built yet")
        M = cfg.MODEL
        num_stage = M.get("num_stage", 4)
        c = M.get("input_channel", 256)
        self.p_drop = float(M.get("ca_dropout", 0.3))
        self.deploy = False
        self.stem = Stem(c, self.p_drop)
        self.backone = EncoderDecoder(num_stage, c, msrb_ca=M.get("msrb_ca", "ca"), rbu_ca=M.get("rbu_ca", "ca"),
                                      p_drop=self.p_drop)
        self.neck = nn.Sequential(RepBasicUnit(c, c, p_drop=self.p_drop), RepBasicUnit(c, c, p_drop=self.p_drop))
        self.head = nn.Conv2d(c, M.get("output_channel", cfg.DATASET.num_joints), 1, 1, 0)
        self.init_weights()

    def emit(self, pb, x, out=None):
        y = self.backone.emit(pb, self.stem.emit(pb, x))
        for m in self.neck:
            y = m.emit(pb, y)
        return pb.conv(y, self.head, None, nchw_out=True)

    def init_weights(self):
        # litehourglass.py:224-230: Conv2d weight ~ N(0,1), bias 0; BatchNorm weight 1, bias 0
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def deploy_model(self):
        """Re-parameterise every unit that can (reference deploy_model; called by test.py:106-107)."""
        for m in self.modules():
            if hasattr(m, "switch_to_deploy"):
                m.switch_to_deploy()
        self.deploy = True
        self.__dict__.pop("_engine", None)

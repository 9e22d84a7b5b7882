"""HIP-backed mirror of models/pose_estimation/lite_hrnet.py (BASELINE config 5: the Lite-HRNet baseline, depth 18 / 30).
Same class attribute names / state_dict keys as the reference (1,483,873 parameters at depth 18).

Everything runs on the litehandnet kernels: the 1x1 / depthwise convolutions take this network's channel counts (20 ... 320,
BatchNorms over 7 / 17 / 37 channels behind outputs padded to a multiple of 4), the elementwise kernels any multiple of 4;
what is new is small: the channel shuffle (`lhn_shuffle2_*`), the product with a nearest-upsampled weight map and the
bilinear upsample-add (`lhn_ew_fwd3` modes + their backward kernels), sigmoid(relu(.)) as a combine activation, the
SpatialWeighting gate as a mode of the squeeze-and-excitation kernels."""
from torch import nn

from .engine import PlanModule
from .plan import EW_BILINEAR, EW_MUL, SLOPE_RELU_SIGMOID


class DWConv(PlanModule):
    """lite_hrnet.py:11-27: depthwise 3x3 + BN [+ ReLU] -> 1x1 + BN [+ ReLU]."""

    def __init__(self, in_channel, out_channel, stride=1, padding=1, dilation=1, mid_relu=True, last_relu=True, bias=False):
        super().__init__()
        self.depthwise_conv = nn.Sequential(
            nn.Conv2d(in_channel, in_channel, 3, stride, padding, groups=in_channel, bias=bias, dilation=dilation),
            nn.BatchNorm2d(in_channel))
        self.mid_relu = nn.ReLU() if mid_relu else nn.Identity()
        self.pointwise_conv = nn.Sequential(nn.Conv2d(in_channel, out_channel, 1, 1, 0, bias=bias), nn.BatchNorm2d(out_channel))
        self.last_relu = nn.ReLU() if last_relu else nn.Identity()

    def emit(self, pb, x, out=None, repeat=1):
        t = pb.conv(x, self.depthwise_conv[0], self.depthwise_conv[1], slope=0.0 if isinstance(self.mid_relu, nn.ReLU) else 1.0,
                    bn_repeat=repeat)
        return pb.conv(t, self.pointwise_conv[0], self.pointwise_conv[1],
                       slope=0.0 if isinstance(self.last_relu, nn.ReLU) else 1.0, out=out, bn_repeat=repeat)


class SpatialWeighting(PlanModule):
    """lite_hrnet.py:55-74: x * sigmoid(relu(conv2(sigmoid(relu(conv1(global_avg_pool(x)))))))  -- a per-(n, c) gate."""

    def __init__(self, channels, ratio=16):
        super().__init__()
        self.global_avgpool = nn.AdaptiveAvgPool2d(1)
        mid_channels = int(channels / ratio)
        self.conv1 = nn.Sequential(nn.Conv2d(channels, mid_channels, 1, 1), nn.ReLU(True), nn.Sigmoid())
        self.conv2 = nn.Sequential(nn.Conv2d(mid_channels, channels, 1, 1), nn.ReLU(True), nn.Sigmoid())

    def emit(self, pb, x, out=None):
        if not pb.owns_buffer(x):
            x = pb.ew([x])
        return pb.se_attention(x, self, convs=(self.conv1[0], self.conv2[0]), mode=1)


class CrossResolutionWeighting(PlanModule):
    """lite_hrnet.py:76-108: every branch pooled to the lowest resolution, concatenated, two 1x1 + BN + ReLU + Sigmoid, split,
    nearest-upsampled and multiplied in."""

    def __init__(self, channels, ratio=16):
        super().__init__()
        self.channels = channels
        total_channel = sum(channels)
        mid_channel = int(total_channel / ratio)
        self.conv1 = nn.Sequential(nn.Conv2d(total_channel, mid_channel, 1, 1), nn.BatchNorm2d(mid_channel), nn.ReLU(True), nn.Sigmoid())
        self.conv2 = nn.Sequential(nn.Conv2d(mid_channel, total_channel, 1, 1), nn.BatchNorm2d(total_channel), nn.ReLU(True), nn.Sigmoid())

    def emit(self, pb, xs, out=None):
        hm, wm = xs[-1].H, xs[-1].W
        pooled = pb.new(hm, wm, sum(self.channels))
        off = 0
        for x in xs:                                   # the pooled branches land side by side: torch.cat without a copy
            pb.avgpool(x, hm, wm, out=pb.slice(pooled, off, x.C))
            off += x.C
        a = pb.ew([pb.conv(pooled, self.conv1[0], self.conv1[1], slope=1.0)], out_slope=SLOPE_RELU_SIGMOID)
        g = pb.ew([pb.conv(a, self.conv2[0], self.conv2[1], slope=1.0)], out_slope=SLOPE_RELU_SIGMOID)
        outs, off = [], 0
        for x in xs:
            outs.append(pb.ew([x, pb.slice(g, off, x.C)], mode=EW_MUL))       # s * F.interpolate(a, size, 'nearest')
            off += x.C
        return outs


class ConditionalChannelWeighting(PlanModule):
    """lite_hrnet.py:110-143."""

    def __init__(self, in_channels, reduce_ratio, stride=1):
        super().__init__()
        branch_channels = [c // 2 for c in in_channels]
        self.cross_resolution_weighting = CrossResolutionWeighting(channels=branch_channels, ratio=reduce_ratio)
        self.depthwise_convs = nn.ModuleList([nn.Sequential(nn.Conv2d(c, c, 3, stride, 1, groups=c), nn.BatchNorm2d(c))
                                              for c in branch_channels])
        self.spatial_weighting = nn.ModuleList([SpatialWeighting(channels=c, ratio=4) for c in branch_channels])

    def emit(self, pb, xs, out=None):
        x1 = [pb.slice(s, 0, s.C // 2) for s in xs]
        x2 = [pb.slice(s, s.C // 2, s.C // 2) for s in xs]
        x2 = self.cross_resolution_weighting.emit(pb, x2)
        x2 = [pb.conv(s, dw[0], dw[1], slope=1.0) for s, dw in zip(x2, self.depthwise_convs)]
        x2 = [sw.emit(pb, s) for s, sw in zip(x2, self.spatial_weighting)]
        return [pb.shuffle2(a, b) for a, b in zip(x1, x2)]


class StageModule(PlanModule):
    """lite_hrnet.py:145-204."""

    def __init__(self, in_branches, num_blocks, in_channels, reduce_ratio=8, with_fuse=True):
        super().__init__()
        self.in_branches = in_branches
        self.in_channels = in_channels
        self.with_fuse = with_fuse
        self.layers = nn.Sequential(*[ConditionalChannelWeighting(in_channels, reduce_ratio) for _ in range(num_blocks)])
        if self.with_fuse and self.in_branches > 1:
            self.fuse_layers = self._make_fuse_layers()
            self.relu = nn.ReLU()
        else:
            self.with_fuse = False

    def _make_fuse_layers(self):
        cs = self.in_channels
        fuse_layers = nn.ModuleList()
        for i in range(self.in_branches):
            fuse_layers.append(nn.ModuleList())
            for j in range(self.in_branches):
                c_in, c_out = cs[j], cs[i]
                if i == j:
                    fuse_layers[-1].append(nn.Identity())
                elif j > i:
                    fuse_layers[-1].append(nn.Sequential(nn.Conv2d(c_in, c_out, 1, 1, 0, bias=False), nn.BatchNorm2d(c_out),
                                                         nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                else:
                    down = [DWConv(c_in, c_in, stride=2, mid_relu=False, last_relu=False) for _ in range(i - j - 1)]
                    down.append(DWConv(c_in, c_out, stride=2, mid_relu=False, last_relu=False))
                    fuse_layers[-1].append(nn.Sequential(*down))
        return fuse_layers

    @staticmethod
    def _sum(pb, terms, out_slope):
        """sum of any number of terms, three per combine (smaller ones nearest-upsampled); the last combine applies out_slope."""
        while len(terms) > 3:
            terms = [pb.ew(terms[:3])] + terms[3:]
        return pb.ew(terms, out_slope=out_slope)

    def _fuse(self, pb, i, j, x, repeat=1):
        m = self.fuse_layers[i][j]
        if j > i:
            return pb.conv(x, m[0], m[1], slope=1.0)             # (the nearest upsample happens in the combine that reads it)
        for d in m:
            x = d.emit(pb, x, repeat=repeat)
        return x

    def emit(self, pb, xs, out=None):
        if self.in_branches == 1:
            return [self.layers[0].emit(pb, xs)[0]]
        out = xs
        for layer in self.layers:
            out = layer.emit(pb, out)
        if not self.with_fuse:
            return out
        nb = self.in_branches
        # lite_hrnet.py:190-199 with its aliasing (see oracle.torch_ref.StageModule): row 0 accumulates INTO out[0], so
        #   y0 = 2*out[0] + sum_j fuse[0][j](out[j])   replaces out[0] for the rows below, which start from fuse[i][0](y0)
        # evaluated twice (value doubled, BatchNorm running statistics moved twice)
        y0 = self._sum(pb, [out[0], out[0]] + [self._fuse(pb, 0, j, out[j]) for j in range(1, nb)], 1.0)
        fused = [pb.ew([y0], out_slope=0.0)]
        for i in range(1, nb):
            v = self._fuse(pb, i, 0, y0, repeat=2)
            terms = [v, v] + [out[j] if j == i else self._fuse(pb, i, j, out[j]) for j in range(1, nb)]
            fused.append(self._sum(pb, terms, 0.0))
        return fused


class StemModule(PlanModule):
    """lite_hrnet.py:206-248."""

    def __init__(self, in_channels, stem_channels, out_channels, expand_ratio):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, stem_channels, 3, 2, 1), nn.BatchNorm2d(stem_channels), nn.ReLU())
        self.out_channels = out_channels
        mid_channels = int(round(stem_channels * expand_ratio))
        branch_channels = stem_channels // 2
        inc_channels = out_channels - branch_channels if stem_channels == out_channels else out_channels - stem_channels
        self.branch1 = DWConv(branch_channels, inc_channels, stride=2, mid_relu=False, bias=True)
        self.expand_conv = nn.Sequential(nn.Conv2d(branch_channels, mid_channels, 1, 1, 0), nn.BatchNorm2d(mid_channels), nn.ReLU())
        self.depthwise_conv = nn.Sequential(nn.Conv2d(mid_channels, mid_channels, 3, 2, 1, groups=mid_channels), nn.BatchNorm2d(mid_channels))
        lc = branch_channels if stem_channels == out_channels else stem_channels
        self.linear_conv = nn.Sequential(nn.Conv2d(mid_channels, lc, 1, 1, 0), nn.BatchNorm2d(lc), nn.ReLU())

    def emit(self, pb, x, out=None):
        x = pb.conv(x, self.conv1[0], self.conv1[1], slope=0.0)
        h = x.C // 2
        x1, x2 = pb.slice(x, 0, h), pb.slice(x, h, h)
        x2 = pb.conv(x2, self.expand_conv[0], self.expand_conv[1], slope=0.0)
        x2 = pb.conv(x2, self.depthwise_conv[0], self.depthwise_conv[1], slope=1.0)
        x2 = pb.conv(x2, self.linear_conv[0], self.linear_conv[1], slope=0.0)
        return pb.shuffle2(self.branch1.emit(pb, x1), x2)


class IterativeHead(PlanModule):
    """lite_hrnet.py:250-281: from the lowest resolution up, bilinear (align_corners) upsample + add, DWConv projection."""

    def __init__(self, in_channels):
        super().__init__()
        num_branches = len(in_channels)
        self.in_channels = in_channels[::-1]
        self.projects = nn.ModuleList([DWConv(self.in_channels[i], self.in_channels[i + 1] if i != num_branches - 1 else self.in_channels[i])
                                       for i in range(num_branches)])

    def emit(self, pb, xs, out=None):
        xs = xs[::-1]
        ys, last = [], None
        for i, s in enumerate(xs):
            if last is not None:
                s = pb.ew([s, last], mode=EW_BILINEAR)
            s = self.projects[i].emit(pb, s)
            ys.append(s)
            last = s
        return ys[::-1]


class LiteHRNet(PlanModule):
    """lite_hrnet.py:284-390.  cfg.MODEL keys: depth (18 | 30), output_channel."""
    consumes_image = True

    def __init__(self, cfg):
        super().__init__()
        out_channel = cfg.MODEL.get("output_channel", cfg.DATASET.num_joints)
        depth = cfg.MODEL.get("depth", 30)
        self.stem = StemModule(in_channels=3, stem_channels=32, out_channels=32, expand_ratio=1)
        self.num_stages = 3
        self.with_head = True
        self.stages_spec = dict(num_modules=(3, 4, 3) if depth == 18 else (3, 8, 3), num_branches=(2, 3, 4), num_blocks=(2, 2, 2),
                                with_fuse=(True, True, True), reduce_ratios=(8, 8, 8),
                                num_channels=((40, 80), (40, 80, 160), (40, 80, 160, 320)))
        num_channels_last = [self.stem.out_channels]
        for i in range(self.num_stages):
            num_channels = list(self.stages_spec["num_channels"][i])
            setattr(self, f"transition{i}", self._make_transition_layer(num_channels_last, num_channels))
            stage, num_channels_last = self._make_stage(self.stages_spec, i, num_channels)
            setattr(self, f"stage{i}", stage)
        self.head_layer = IterativeHead(in_channels=num_channels_last)
        self.out_conv = nn.Conv2d(40, out_channel, 1, 1, 0)

    @staticmethod
    def _make_transition_layer(pre, cur):
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                layers.append(DWConv(pre[i], cur[i], mid_relu=False) if cur[i] != pre[i] else None)
            else:
                down = []
                for j in range(i + 1 - len(pre)):
                    c_in = pre[-1]
                    down.append(DWConv(c_in, cur[i] if j == i - len(pre) else c_in, stride=2, mid_relu=False))
                layers.append(nn.Sequential(*down))
        return nn.ModuleList(layers)

    @staticmethod
    def _make_stage(spec, si, in_channels):
        modules = []
        for _ in range(spec["num_modules"][si]):
            modules.append(StageModule(spec["num_branches"][si], spec["num_blocks"][si], in_channels, spec["reduce_ratios"][si],
                                       spec["with_fuse"][si]))
            in_channels = modules[-1].in_channels
        return nn.Sequential(*modules), in_channels

    def emit(self, pb, x, out=None):
        y_list = [self.stem.emit(pb, x)]
        for i in range(self.num_stages):
            transition = getattr(self, f"transition{i}")
            x_list = []
            for j in range(self.stages_spec["num_branches"][i]):
                t = transition[j]
                if t is None:
                    x_list.append(y_list[j])
                    continue
                src = y_list[-1] if j >= len(y_list) else y_list[j]
                if isinstance(t, DWConv):
                    x_list.append(t.emit(pb, src))
                else:
                    for d in t:
                        src = d.emit(pb, src)
                    x_list.append(src)
            y_list = x_list
            for module in getattr(self, f"stage{i}"):
                y_list = module.emit(pb, y_list)
        y_list = self.head_layer.emit(pb, y_list)
        return pb.conv(y_list[0], self.out_conv, None, nchw_out=True)

"""HIP-backed mirror of models/pose_estimation/hourglassnet.py (BASELINE config 5: the stacked-hourglass baseline the
reference compares litehandnet with).  Same class attribute names / state_dict keys; output [N, num_stack, K, H/4, W/4].

Kernel reuse: every convolution is one of the litehandnet kernels -- the 1x1s (64..256 channels, run by the library as
128-wide slices), the dense 3x3 implicit GEMM (64->64, 128->128) and the 3-channel stem (7x7 stride 2).  The pre-activation
unit BN -> ReLU -> conv costs no pass of its own except for bn1, whose input is a plain sum: a BatchNorm in front of a
convolution is the PENDING transform of that convolution's input (`pb.conv(x, conv, bn_of_the_next_unit)`)."""
from torch import nn

from .engine import PlanModule

Pool = nn.MaxPool2d


class Conv(PlanModule):
    """hourglassnet.py:6-25."""

    def __init__(self, inp_dim, out_dim, kernel_size=3, stride=1, bn=False, relu=True):
        super().__init__()
        self.inp_dim = inp_dim
        self.conv = nn.Conv2d(inp_dim, out_dim, kernel_size, stride, padding=(kernel_size - 1) // 2, bias=True)
        self.relu = nn.ReLU() if relu else None
        self.bn = nn.BatchNorm2d(out_dim) if bn else None

    def emit(self, pb, x, out=None):
        return pb.conv(x, self.conv, self.bn, slope=0.0 if self.relu is not None else 1.0, out=out)


class Residual(PlanModule):
    """hourglassnet.py:27-53."""

    def __init__(self, inp_dim, out_dim):
        super().__init__()
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm2d(inp_dim)
        self.conv1 = Conv(inp_dim, int(out_dim / 2), 1, relu=False)
        self.bn2 = nn.BatchNorm2d(int(out_dim / 2))
        self.conv2 = Conv(int(out_dim / 2), int(out_dim / 2), 3, relu=False)
        self.bn3 = nn.BatchNorm2d(int(out_dim / 2))
        self.conv3 = Conv(int(out_dim / 2), out_dim, 1, relu=False)
        self.skip_layer = nn.Identity() if inp_dim == out_dim else Conv(inp_dim, out_dim, 1, relu=False)

    def emit(self, pb, x, out=None):
        res = x if isinstance(self.skip_layer, nn.Identity) else self.skip_layer.emit(pb, x)
        t = pb.bn_only(x, self.bn1, slope=0.0)                              # relu(bn1(x)): statistics pass, pending transform
        t = pb.conv(t, self.conv1.conv, self.bn2, slope=0.0)                # relu(bn2(conv1(.)))
        t = pb.conv(t, self.conv2.conv, self.bn3, slope=0.0)                # relu(bn3(conv2(.)))
        t = pb.conv(t, self.conv3.conv, None)
        return pb.ew([t, res], out=out)


class HourglassModule(PlanModule):
    """hourglassnet.py:55-81."""

    def __init__(self, n, f, bn=None, increase=0):
        super().__init__()
        nf = f + increase
        self.up1 = Residual(f, f)
        self.pool1 = Pool(2, 2)
        self.low1 = Residual(f, nf)
        self.n = n
        self.low2 = HourglassModule(n - 1, nf, bn=bn) if n > 1 else Residual(nf, nf)
        self.low3 = Residual(nf, f)
        self.up2 = nn.Upsample(scale_factor=2, mode="nearest")

    def emit(self, pb, x, out=None):
        up1 = self.up1.emit(pb, x)
        low = self.low3.emit(pb, self.low2.emit(pb, self.low1.emit(pb, pb.maxpool(x))))
        return pb.ew([up1, low], out=out)                                   # nearest x2 upsample + add in one pass


class Merge(PlanModule):
    def __init__(self, x_dim, y_dim):
        super().__init__()
        self.conv = Conv(x_dim, y_dim, 1, relu=False, bn=False)

    def emit(self, pb, x, out=None):
        return self.conv.emit(pb, x, out)


class HourglassNet(PlanModule):
    """hourglassnet.py:90-136.  cfg.MODEL keys: num_stack, num_level, input_channel, output_channel."""
    consumes_image = True
    stacked_output = True

    def __init__(self, cfg):
        super().__init__()
        M = cfg.MODEL
        num_stack, num_level = M.get("num_stack", 8), M.get("num_level", 4)
        inp_dim, oup_dim = M.get("input_channel", 256), M.get("output_channel", 21)
        self.num_stack = num_stack
        self.pre = nn.Sequential(Conv(3, 64, 7, 2, bn=True, relu=True), Residual(64, 128), Pool(2, 2), Residual(128, 128),
                                 Residual(128, inp_dim))
        self.hgs = nn.ModuleList([nn.Sequential(HourglassModule(num_level, inp_dim, bn=False, increase=0))
                                  for _ in range(num_stack)])
        self.features = nn.ModuleList([nn.Sequential(Residual(inp_dim, inp_dim), Conv(inp_dim, inp_dim, 1, bn=True, relu=True))
                                       for _ in range(num_stack)])
        self.outs = nn.ModuleList([Conv(inp_dim, oup_dim, 1, relu=False, bn=False) for _ in range(num_stack)])
        self.merge_features = nn.ModuleList([Merge(inp_dim, inp_dim) for _ in range(num_stack - 1)])
        self.merge_preds = nn.ModuleList([Merge(oup_dim, inp_dim) for _ in range(num_stack - 1)])

    def emit(self, pb, x, out=None):
        for m in self.pre:
            x = pb.maxpool(x) if isinstance(m, nn.MaxPool2d) else m.emit(pb, x)
        S, y = self.num_stack, None
        for i in range(S):
            f = self.hgs[i][0].emit(pb, x)
            for m in self.features[i]:
                f = m.emit(pb, f)
            y = pb.conv(f, self.outs[i].conv, None, nchw_out=True, stack=(i, S))        # slot i of [N, S, K, H, W]
            if i < S - 1:
                # the predictions feed the next stack as well: a second, NHWC copy (K padded to a multiple of 4 with exact
                # zeros) -- both head convolutions add into the same weight gradient
                p = pb.conv(f, self.outs[i].conv, None)
                x = pb.ew([x, self.merge_preds[i].emit(pb, p), self.merge_features[i].emit(pb, f)])
        return y

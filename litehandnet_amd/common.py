"""HIP-backed mirror of models/pose_estimation/liteHandNet/common.py:23-66 (SEBlock, ChannelAttension)."""
from torch import nn

from .engine import PlanModule


class SEBlock(PlanModule):
    """x * sigmoid(up(relu(down(global_avg_pool(x)))))  -- common.py:23-37."""

    def __init__(self, input_channels, internal_neurons):
        super().__init__()
        self.down = nn.Conv2d(input_channels, internal_neurons, kernel_size=1, stride=1, bias=True)
        self.up = nn.Conv2d(internal_neurons, input_channels, kernel_size=1, stride=1, bias=True)
        self.input_channels = input_channels

    def emit(self, pb, x, out=None):
        if not pb.owns_buffer(x):
            x = pb.ew([x])          # a gate needs a buffer of its own whose producer turns (gate, dz, dpool) into d(raw)
        return pb.se_attention(x, self)


class ChannelAttension(PlanModule):
    """x * sigmoid(W2 lrelu(W1 dropout2d(BN(dw3x3_valid(adaptive_avg_pool(x,3))))))  (the reference's spelling)."""

    def __init__(self, channel, deploy=False, p_drop=0.3):
        super().__init__()
        self.deploy = deploy
        if deploy:
            self.rbr_reparam = nn.Conv2d(channel, channel, 3, 1, 0, groups=channel)
        else:
            self.conv3x3 = nn.Sequential()
            self.conv3x3.add_module("conv", nn.Conv2d(channel, channel, 3, 1, 0, groups=channel, bias=False))
            self.conv3x3.add_module("bn", nn.BatchNorm2d(channel))
        self.conv1x1 = nn.Sequential(nn.Dropout2d(p=p_drop), nn.Conv2d(channel, channel // 2, 1, 1, 0),
                                     nn.LeakyReLU(inplace=True), nn.Conv2d(channel // 2, channel, 1, 1, 0),
                                     nn.Sigmoid())

    def switch_to_deploy(self):
        """common.py:68-90: fold the BatchNorm of the 3x3 depthwise conv into a biased conv."""
        if hasattr(self, "rbr_reparam"):
            return
        from .repblocks import fused_conv
        self.rbr_reparam = fused_conv(self.conv3x3.conv, [(self.conv3x3.conv.weight, self.conv3x3.bn)])
        del self.conv3x3
        self.deploy = True

    def emit(self, pb, x, out=None):
        # the gate attaches to the buffer behind x; a view that does not own its buffer is materialised first
        if not pb.owns_buffer(x):
            x = pb.ew([x])
        return pb.channel_attention(x, self)

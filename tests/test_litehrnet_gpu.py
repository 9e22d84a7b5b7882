"""BASELINE config 5, part 2: the Lite-HRNet baseline (models/pose_estimation/lite_hrnet.py) on the litehandnet kernels:
odd channel counts (20 ... 320, BatchNorms over 7 / 17 / 37 channels), channel shuffle, the cross-resolution product,
the bilinear head, fuse layers with the reference's aliasing -- blocks against the float64 oracle, the whole network
against the real reference's vectors (tests/golden/make_golden_r2_models.py)."""
import pytest
import torch
import torch.nn.functional as F

from litehandnet_amd.config import litehandnet_cfg
from oracle import synth, torch_ref
from test_model_gpu import _check_block, _model_case, _x

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,cout", [(20, 20), (40, 80), (80, 40), (160, 96), (320, 40), (300, 60), (32, 40)])
def test_pointwise_odd_channels(dev, cin, cout):
    """1x1 + BatchNorm + LeakyReLU for channel counts that are multiples of 4 but not of 32: padded K tiles, sliced wide inputs."""
    from litehandnet_amd import repblocks
    _check_block(repblocks.RepConv(cin, cout, 1), torch_ref.RepConv(cin, cout, 1), _x(3, cin, 12, 20, seed=cin + cout), dev, seed=cin)


@pytest.mark.parametrize("c,stride", [(20, 1), (40, 2), (80, 1), (160, 1), (320, 1), (16, 2)])
def test_depthwise_odd_channels(dev, c, stride):
    from litehandnet_amd import lite_hrnet as lh
    _check_block(lh.DWConv(c, c, stride=stride, mid_relu=False, bias=(c == 16)),
                 torch_ref.LHDWConv(c, c, stride=stride, mid_relu=False, bias=(c == 16)), _x(3, c, 16, 16, seed=c), dev, seed=c + 1)


def test_spatial_weighting(dev):
    from litehandnet_amd import lite_hrnet as lh
    _check_block(lh.SpatialWeighting(40, 4), torch_ref.SpatialWeighting(40, 4), _x(4, 40, 12, 12, seed=3), dev, seed=4)
    _check_block(lh.SpatialWeighting(160, 4), torch_ref.SpatialWeighting(160, 4), _x(4, 160, 8, 8, seed=5), dev, seed=6)


# ---- list-in / list-out blocks behind single-tensor wrappers (same construction on both sides): branch 1 = two copies of the
# 2x2 max-pooled input side by side, output = branch 0 + nearest-upsampled first half of branch 1
def _wrap_pair(block_ours, block_ref):
    from litehandnet_amd.engine import PlanModule

    class Ours(PlanModule):
        def __init__(self):
            super().__init__()
            self.m = block_ours

        def emit(self, pb, x, out=None):
            p = pb.maxpool(x)
            o = self.m.emit(pb, [x, pb.cat([p, p])])
            return pb.ew([o[0], pb.slice(o[1], 0, x.C)])

    class Ref(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.m = block_ref

        def forward(self, x):
            p = F.max_pool2d(x, 2, 2)
            o = self.m([x, torch.cat([p, p], 1)])
            return o[0] + F.interpolate(o[1][:, :x.shape[1]], size=x.shape[-2:], mode="nearest")
    return Ours(), Ref()


def test_conditional_channel_weighting(dev):
    from litehandnet_amd import lite_hrnet as lh
    ours, ref = _wrap_pair(lh.ConditionalChannelWeighting([40, 80], 8), torch_ref.ConditionalChannelWeighting([40, 80], 8))
    _check_block(ours, ref, _x(4, 40, 16, 16, seed=7), dev, seed=8)


def test_stage_module_fuse_aliasing(dev):
    """StageModule with its fuse layers: row 0 accumulates into out[0] (lite_hrnet.py:192-197), the rows below read that
    sum through a down-sampling unit evaluated twice -- value doubled, BatchNorm running statistics and num_batches_tracked
    moved twice per forward (checked by _check_block against the oracle, which is pinned to the reference)."""
    from litehandnet_amd import lite_hrnet as lh
    ours, ref = _wrap_pair(lh.StageModule(2, 2, [40, 80], 8, True), torch_ref.StageModule(2, 2, [40, 80], 8, True))
    _check_block(ours, ref, _x(4, 40, 16, 16, seed=9), dev, seed=10, grad_tol=5e-3)


def test_iterative_head_bilinear(dev):
    from litehandnet_amd import lite_hrnet as lh
    from litehandnet_amd.engine import PlanModule

    class Ours(PlanModule):
        def __init__(self):
            super().__init__()
            self.m = lh.IterativeHead([40, 80])

        def emit(self, pb, x, out=None):
            p = pb.maxpool(x)
            return self.m.emit(pb, [x, pb.single(pb.cat([p, p]))])[0]

    class Ref(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.m = torch_ref.IterativeHead([40, 80])

        def forward(self, x):
            p = F.max_pool2d(x, 2, 2)
            return self.m([x, torch.cat([p, p], 1)])[0]
    _check_block(Ours(), Ref(), _x(3, 40, 12, 20, seed=11), dev, seed=12)


def test_stem(dev):
    from litehandnet_amd import lite_hrnet as lh

    class Stem(lh.StemModule):
        consumes_image = True          # the block reads the NCHW image itself (3-channel stem kernel), like the full model
    _check_block(Stem(3, 32, 32, 1), torch_ref.LHStemModule(3, 32, 32, 1), _x(2, 3, 64, 64, seed=13), dev, seed=14, no_dx=True)


def test_state_dict_contract_litehrnet(dev):
    from litehandnet_amd import get_model
    for depth, want in ((18, 1483873), (30, 1773361)):            # test_models_performance.ipynb:276-279 (depth 18)
        cfg = litehandnet_cfg("L", depth=depth)
        ours, ref = get_model(cfg), torch_ref.get_model(cfg)
        assert list(ours.state_dict()) == list(ref.state_dict())
        assert all(a.shape == b.shape for a, b in zip(ours.state_dict().values(), ref.state_dict().values()))
        assert sum(p.numel() for p in ours.parameters()) == want


def test_model_L18_128_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "L18_128", variant="L", depth=18)


def test_model_L18_256_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "L18_256", variant="L", depth=18)


def test_model_L30_128_golden(dev, golden_dir):
    """Lite-HRNet-30 (config/litehrnet/_1_*_30.py: 3 / 8 / 3 modules per stage) against the reference's fixture."""
    _model_case(dev, golden_dir, "L30_128", variant="L", depth=30)


def test_litehrnet_whole_model_gradients_elementwise(dev):
    """Every parameter gradient of the whole Lite-HRNet-18 element by element against the float64 oracle (the golden cases compare
    gradient norms): cross-branch routing, the twice-evaluated fuse layers, the iterative head (lite_hrnet.py:145-282)."""
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg("L", depth=18)
    cfg.MODEL["ca_dropout"] = 0.0
    ours, ref = get_model(cfg), torch_ref.get_model(cfg)
    _check_block(ours, ref, synth.synth_images(2, 128, 63), dev, seed=64, no_dx=True, grad_tol=2e-2)


def test_litehrnet_eval(dev):
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg("L", depth=18)
    ours, ref = get_model(cfg), torch_ref.get_model(cfg)
    sd = synth.synth_state_dict(ref, 21)
    ref.load_state_dict(sd); ours.load_state_dict(sd)
    ours.to(dev).eval(); ref.double().eval()
    x = synth.synth_images(3, 128, 5)
    with torch.no_grad():
        y, y64 = ours(x.to(dev)), ref(x.double())
    e = float((y.cpu().double() - y64).abs().max() / y64.abs().max())
    assert e < 1e-4, e

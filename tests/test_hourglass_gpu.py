"""BASELINE config 5, part 1: the stacked-hourglass baseline (models/pose_estimation/hourglassnet.py) on the litehandnet
kernels -- 1x1 convolutions up to 256 channels (run as 128-wide slices by the library), the 7x7 stem, the pre-activation
unit; forward + backward against the float64 oracle and the real reference's vectors (tests/golden/make_golden_r2.py)."""
import numpy as np
import pytest
import torch

from litehandnet_amd.config import litehandnet_cfg
from oracle import synth, torch_ref
from test_model_gpu import _check_block, _model_case, _x

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,cout", [(256, 256), (256, 128), (128, 256), (256, 64), (64, 256), (256, 32)])
def test_pointwise_wide_and_odd_channels(dev, cin, cout):
    """RepConv 1x1 + BatchNorm + LeakyReLU with channel counts beyond one 128-wide tile: input slices accumulate into y,
    statistics come from the last slice, the backward takes the split (dgrad / wgrad) or the fused sliced path."""
    from litehandnet_amd import repblocks
    _check_block(repblocks.RepConv(cin, cout, 1), torch_ref.RepConv(cin, cout, 1), _x(3, cin, 12, 20, seed=cin + cout), dev, seed=cin)


def test_biased_pointwise_wide(dev):
    """hourglassnet.py Conv(bn=False, relu=False): plain biased 1x1 (conv3 / skip / merge layers), 128 -> 256 and 256 -> 256,
    incl. d(bias); and Conv(256, 256, 1, bn=True, relu=True)."""
    from litehandnet_amd import hourglassnet as hg
    _check_block(hg.Conv(128, 256, 1, relu=False), torch_ref.HGConv(128, 256, 1, relu=False), _x(3, 128, 8, 12, seed=1), dev, seed=2)
    _check_block(hg.Conv(256, 256, 1, relu=False), torch_ref.HGConv(256, 256, 1, relu=False), _x(3, 256, 8, 12, seed=3), dev, seed=4)
    _check_block(hg.Conv(256, 256, 1, bn=True, relu=True), torch_ref.HGConv(256, 256, 1, bn=True, relu=True), _x(3, 256, 8, 12, seed=5), dev, seed=6)


@pytest.mark.parametrize("cin,cout", [(64, 128), (128, 128), (128, 256), (256, 256)])
def test_residual(dev, cin, cout):
    from litehandnet_amd import hourglassnet as hg
    _check_block(hg.Residual(cin, cout), torch_ref.HGResidual(cin, cout), _x(4, cin, 8, 8, seed=cin), dev, seed=cout)


def test_hourglass_module(dev):
    from litehandnet_amd import hourglassnet as hg
    _check_block(hg.HourglassModule(3, 128), torch_ref.HourglassModule(3, 128), _x(2, 128, 16, 16, seed=9), dev, seed=10)
    _check_block(hg.HourglassModule(2, 256), torch_ref.HourglassModule(2, 256), _x(2, 256, 8, 8, seed=11), dev, seed=12)


def test_stem_7x7(dev):
    from litehandnet_amd import hourglassnet as hg

    class Pre(hg.PlanModule):
        consumes_image = True

        def __init__(self):
            super().__init__()
            self.c = hg.Conv(3, 64, 7, 2, bn=True, relu=True)

        def emit(self, pb, x, out=None):
            return self.c.emit(pb, x)

    class PreRef(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch_ref.HGConv(3, 64, 7, 2, bn=True, relu=True)

        def forward(self, x):
            return self.c(x)
    _check_block(Pre(), PreRef(), _x(2, 3, 40, 48, seed=13), dev, seed=14, no_dx=True)


def test_state_dict_contract_hourglass(dev):
    from litehandnet_amd import get_model
    for ns, want in ((1, 3427733), (2, 6574250)):              # debug_litehandnet.ipynb:542 (1 stack)
        cfg = litehandnet_cfg("H", num_stack=ns)
        ours, ref = get_model(cfg), torch_ref.get_model(cfg)
        assert list(ours.state_dict()) == list(ref.state_dict())
        assert all(a.shape == b.shape for a, b in zip(ours.state_dict().values(), ref.state_dict().values()))
        assert sum(p.numel() for p in ours.parameters()) == want


def test_model_H1_128_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "H1_128", variant="H", num_stack=1)


def test_model_H1_256_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "H1_256", variant="H", num_stack=1)


def test_model_H2_128_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "H2_128", variant="H", num_stack=2)


def test_model_H2_256_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "H2_256", variant="H", num_stack=2)


def test_hourglass_whole_model_gradients_elementwise(dev):
    """Every parameter gradient of the whole 2-stack network ELEMENT BY ELEMENT against the float64 oracle (the golden cases above
    compare gradient norms only): stack wiring, the padded 24-channel prediction copy feeding merge_preds, the two head
    convolutions, the stacked [N, S, K, H, W] output gradient (hourglassnet.py:124-136)."""
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg("H", num_stack=2)
    ours, ref = get_model(cfg), torch_ref.get_model(cfg)
    _check_block(ours, ref, synth.synth_images(2, 128, 61), dev, seed=62, no_dx=True, grad_tol=2e-2)


def test_hourglass_eval_and_output_shape(dev):
    """eval mode (running statistics) against the float64 oracle; output is [N, num_stack, K, H/4, W/4] even for one stack."""
    from litehandnet_amd import get_model
    for ns in (1, 2):
        cfg = litehandnet_cfg("H", num_stack=ns)
        ours, ref = get_model(cfg), torch_ref.get_model(cfg)
        sd = synth.synth_state_dict(ref, 20 + ns)
        ref.load_state_dict(sd); ours.load_state_dict(sd)
        ours.to(dev).eval(); ref.double().eval()
        x = synth.synth_images(3, 128, 5)
        with torch.no_grad():
            y, y64 = ours(x.to(dev)), ref(x.double())
        assert tuple(y.shape) == (3, ns, 21, 32, 32) == tuple(y64.shape)
        e = float((y.cpu().double() - y64).abs().max() / y64.abs().max())
        assert e < 1e-4, e

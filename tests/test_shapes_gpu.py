"""GPU parity over awkward shapes: odd / tiny / non-square maps, N = 1..3, every convolution kind the plan emits
(1x1 with and without stride, depthwise 3x3 dil 1/2 and stride 2, 7x7 depthwise, dense 3x3 stride 1/2, the three-branch
RepBlock) -- partial GEMM tiles, partial LDS tiles, gather fallbacks (W < 8) and the parity dgrad path all get hit.
Arbiter and bars as in test_model_gpu._check_block (oracle in float64)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref
from test_model_gpu import _check_block, _x

pytestmark = pytest.mark.gpu

CASES = [
    # (cin, cout, k, stride, pad, dil, groups, act, N, H, W)
    (64, 64, 1, 1, 0, 1, 1, "lrelu", 1, 5, 7),
    (32, 128, 1, 1, 0, 1, 1, None, 3, 9, 13),
    (128, 64, 1, 2, 0, 1, 1, None, 2, 11, 11),
    (128, 128, 1, 1, 0, 1, 1, "lrelu", 2, 7, 9),
    (64, 64, 3, 1, 1, 1, 64, None, 2, 7, 7),
    (64, 64, 3, 1, 2, 2, 64, None, 3, 9, 20),
    (32, 32, 3, 1, 2, 2, 32, "lrelu", 1, 17, 33),
    (32, 32, 3, 2, 1, 1, 32, None, 2, 15, 15),
    (128, 128, 3, 1, 1, 1, 128, "lrelu", 2, 8, 40),
    (20, 20, 3, 1, 1, 1, 20, None, 2, 9, 11),             # channel tails of the tiled depthwise kernels (C % 32 != 0)
    (40, 40, 3, 1, 2, 2, 40, "lrelu", 1, 17, 33),
    (72, 72, 3, 1, 1, 1, 72, "lrelu", 3, 8, 8),
    (36, 36, 3, 1, 2, 2, 36, None, 2, 10, 12),
    (20, 20, 3, 2, 1, 1, 20, None, 2, 9, 11),              # stride-2 tiled depthwise: odd maps, channel tail
    (64, 64, 3, 2, 1, 1, 64, "lrelu", 3, 8, 34),
    (40, 40, 3, 2, 1, 1, 40, None, 1, 33, 18),
    (32, 32, 3, 1, 1, 1, 1, "lrelu", 3, 6, 10),
    (64, 64, 3, 1, 1, 1, 1, None, 2, 9, 9),
    (128, 128, 3, 1, 1, 1, 1, "lrelu", 1, 5, 5),
    (128, 128, 3, 2, 1, 1, 1, "lrelu", 2, 14, 10),
    (64, 64, 3, 2, 1, 1, 1, None, 3, 7, 7),
    (32, 32, 3, 2, 1, 1, 1, "lrelu", 2, 9, 13),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_repconv_shapes(dev, case):
    from litehandnet_amd import repblocks
    cin, cout, k, stride, pad, dil, groups, act, n, h, w = case
    a = torch.nn.LeakyReLU if act else None
    ours = repblocks.RepConv(cin, cout, k, stride, pad, dil, groups, activation=a)
    ref = torch_ref.RepConv(cin, cout, k, stride, pad, dil, groups, activation=a)
    _check_block(ours, ref, _x(n, cin, h, w, seed=h * 31 + w), dev, seed=cin + k)


@pytest.mark.parametrize("n,h,w", [(1, 9, 9), (2, 6, 20), (3, 13, 7)])
def test_repblock_shapes(dev, n, h, w):
    from litehandnet_amd import repblocks
    _check_block(repblocks.RepBlock(32, 32, 7, 1, 3, groups=32), torch_ref.RepBlock(32, 32, 7, 1, 3, groups=32),
                 _x(n, 32, h, w, seed=5), dev, seed=9)
    _check_block(repblocks.RepBlock(32, 64, 3, 1, 1), torch_ref.RepBlock(32, 64, 3, 1, 1), _x(n, 32, h, w, seed=6), dev, seed=10)

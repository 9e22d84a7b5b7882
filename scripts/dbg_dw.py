import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth, torch_ref
from litehandnet_amd import litehourglass as lh, repblocks
exec(open(os.path.join(os.path.dirname(__file__), "dbg_hg.py")).read().split("for (n, c, s, st) in")[0])
for (c, dil, n, h, w) in [(64, 1, 2, 8, 8), (64, 1, 2, 16, 16), (32, 1, 2, 8, 8), (64, 2, 2, 8, 8), (64, 1, 2, 8, 16), (64, 1, 2, 16, 8), (128, 1, 2, 12, 20)]:
    print("dw", c, dil, n, h, w, end=": ")
    run(repblocks.RepConv(c, c, 3, 1, dil, dil, groups=c, activation=None), torch_ref.RepConv(c, c, 3, 1, dil, dil, groups=c, activation=None),
        torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)))
for (c, n, s) in [(128, 2, 16), (128, 2, 8)]:
    print("MSRB ca", c, n, s, end=": ")
    run(lh.MSRB(c, c, "ca", p_drop=0.0), torch_ref.MSRB(c, c, "ca", 0.0), torch.randn(n, c, s, s, generator=torch.Generator().manual_seed(0)))
    print("RBU none", c, n, s, end=": ")
    run(lh.RepBasicUnit(c, c, "none", p_drop=0.0), torch_ref.RepBasicUnit(c, c, "none", 0.0), torch.randn(n, c, s, s, generator=torch.Generator().manual_seed(0)))

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth, torch_ref
from litehandnet_amd import litehourglass as lh
sys.argv = sys.argv[:1]
exec(open(os.path.join(os.path.dirname(__file__), "dbg_hg.py")).read().split("for (n, c, s, st) in")[0])
for st in (1, 2):
    for ca in ("none", "ca"):
        for s in (16, 32, 64):
            print("stages", st, "msrb_ca", ca, "size", s, end=": ")
            run(lh.EncoderDecoder(st, 64, ca, "none", p_drop=0.0), torch_ref._HourglassB(st, 64, ca, "none", 0.0),
                torch.randn(2, 64, s, s, generator=torch.Generator().manual_seed(0)), pick=-1)

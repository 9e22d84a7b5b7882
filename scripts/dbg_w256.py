"""Bisect helper: variant-B blocks at C = 256 against the float64 oracle (forward only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from litehandnet_amd import litehourglass as lh
from oracle import synth, torch_ref
dev = torch.device("cuda:0")
def x(n, c, h, w, seed=0):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal((n, c, h, w)).astype(np.float32))
def run(name, ours, ref, inp, pick=None):
    sd = synth.synth_state_dict(ref, 1); ref.load_state_dict(sd); ours.load_state_dict(sd)
    ours.to(dev).train(); ref.double().train()
    try:
        with torch.no_grad():
            y = ours(inp.to(dev)); yr = ref(inp.double())
        if pick is not None: yr = yr[pick]
        e = float((y.cpu().double() - yr).abs().max() / yr.abs().max())
        print(f"{name}: rel err {e:.2e}", flush=True)
    except Exception as ex:
        print(f"{name}: EXC {ex!r}", flush=True)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
run("rbu none", lh.RepBasicUnit(C, C, "none", p_drop=0.0), torch_ref.RepBasicUnit(C, C, "none", 0.0), x(4, C, 16, 16))
run("rbu ca", lh.RepBasicUnit(C, C, "ca", p_drop=0.0), torch_ref.RepBasicUnit(C, C, "ca", 0.0), x(4, C, 16, 16))
run("msrb none", lh.MSRB(C, C, "none", p_drop=0.0), torch_ref.MSRB(C, C, "none", 0.0), x(4, C, 16, 16))
run("msrb ca", lh.MSRB(C, C, "ca", p_drop=0.0), torch_ref.MSRB(C, C, "ca", 0.0), x(4, C, 16, 16))
run("stem", lh.Stem(C, p_drop=0.0), torch_ref._StemB(C, 0.0), x(2, 3, 64, 64))
run("encdec", lh.EncoderDecoder(4, C, "ca", "none", p_drop=0.0), torch_ref._HourglassB(4, C, "ca", "none", 0.0), x(4, C, 32, 32), pick=-1)
from litehandnet_amd.common import ChannelAttension
run("ca", ChannelAttension(C, p_drop=0.0), torch_ref.ChannelAttension(C, 0.0), x(8, C, 12, 12))
from litehandnet_amd import repblocks
run("dw3 d1", repblocks.RepConv(C // 2, C // 2, 3, 1, 1, groups=C // 2, activation=None), torch_ref.RepConv(C // 2, C // 2, 3, 1, 1, groups=C // 2, activation=None), x(2, C // 2, 16, 16))
run("dw3 d2", repblocks.RepConv(C // 2, C // 2, 3, 1, 2, 2, groups=C // 2, activation=None), torch_ref.RepConv(C // 2, C // 2, 3, 1, 2, 2, groups=C // 2, activation=None), x(2, C // 2, 16, 16))
run("dw3 C", repblocks.RepConv(C, C, 3, 1, 1, groups=C), torch_ref.RepConv(C, C, 3, 1, 1, groups=C), x(2, C, 16, 16))

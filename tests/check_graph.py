"""Runs a few training iterations of variant B on a side stream and prints a checksum; used by tests to compare
LHN_GRAPH=1 (hipGraph replay of the plan) with plain launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litehandnet_amd import get_model
from litehandnet_amd.config import litehandnet_cfg
from oracle import synth

cfg = litehandnet_cfg("B")
cfg.MODEL["ca_dropout"] = 0.0
m = get_model(cfg)
m.load_state_dict(synth.synth_state_dict(m, 3))
m.cuda().train()
x = synth.synth_images(4, 64, 1).cuda()
g = torch.ones(4, 21, 16, 16, device="cuda")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
outs = []
with torch.cuda.stream(side):
    for it in range(5):
        m.zero_grad()
        y = m(x)
        y.backward(g)
        outs.append((float(y.double().abs().sum()), float(sum(p.grad.double().abs().sum() for p in m.parameters()))))
torch.cuda.synchronize()
print("CHECK", " ".join(f"{a:.10e}/{b:.10e}" for a, b in outs))

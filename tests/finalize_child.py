"""Child process of test_fused_finalize_agrees_with_separate_finalize: two training steps of variant B on seeded inputs, outputs /
loss / gradients / running statistics written to an .npz.  LHN_FUSE_FINALIZE is read once per process, hence the child.

    python tests/finalize_child.py OUT.npz"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from litehandnet_amd import get_loss, get_model  # noqa: E402
from litehandnet_amd.config import litehandnet_cfg  # noqa: E402
from oracle import synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = litehandnet_cfg("B")
cfg.MODEL["ca_dropout"] = 0.0
crit = get_loss(cfg)
m = get_model(cfg)
m.load_state_dict({k: v.clone() for k, v in synth.synth_state_dict(m, 5).items()})
m.to(dev).train()
x = synth.synth_images(4, 128, 7).to(dev)
t = torch.rand(4, 21, 32, 32, generator=torch.Generator().manual_seed(3)).to(dev)
meta = {"target": t, "target_weight": torch.ones(4, 21, 1, device=dev)}
out = {}
for step in range(2):                      # the second step starts from running statistics the first one moved
    y = m(x)
    loss, _ = crit(y, meta)
    m.zero_grad()
    loss.backward()
    out[f"y{step}"] = y.detach().cpu().numpy()
    out[f"loss{step}"] = np.float64(float(loss))
    for k, p in m.named_parameters():
        out[f"g{step}.{k}"] = p.grad.detach().cpu().numpy()
for k, b in m.named_buffers():
    out[f"b.{k}"] = b.detach().cpu().numpy()
np.savez(sys.argv[1], **out)

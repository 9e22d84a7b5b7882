"""Profiling driver: a few training steps + forward-only passes of variant B/A at bs64 (run under rocprofv3)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litehandnet_amd import get_loss, get_model, heatmap
from litehandnet_amd.config import litehandnet_cfg
from litehandnet_amd.train import Trainer
ap = argparse.ArgumentParser(); ap.add_argument("--variant", default="B"); ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--fwd-only", action="store_true"); a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = litehandnet_cfg(a.variant)
m = get_model(cfg).to(dev).train(); crit = get_loss(cfg)
img = torch.randn(a.batch, 3, 256, 256, device=dev)
j = torch.zeros(a.batch, 21, 3, device=dev); j[..., :2] = torch.rand(a.batch, 21, 2, device=dev) * 256
t, w = heatmap.generate_target_batch(j, torch.ones_like(j), [256, 256], [64, 64], 2, True)
if a.fwd_only:
    with torch.no_grad():
        for _ in range(a.steps): m(img)
else:
    tr = Trainer(m, crit)
    for _ in range(a.steps): tr.step(img, {"target": t, "target_weight": w})
torch.cuda.synchronize()
print("done")

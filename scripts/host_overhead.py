"""Host-side enqueue time per training step (how far ahead of the GPU the CPU runs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litehandnet_amd import get_loss, get_model, heatmap
from litehandnet_amd.config import litehandnet_cfg
from litehandnet_amd.train import Trainer
v = sys.argv[1] if len(sys.argv) > 1 else "B"
dev = torch.device("cuda:0")
cfg = litehandnet_cfg(v)
m = get_model(cfg).to(dev).train(); tr = Trainer(m, get_loss(cfg))
img = torch.randn(64, 3, 256, 256, device=dev)
j = torch.zeros(64, 21, 3, device=dev); j[..., :2] = torch.rand(64, 21, 2, device=dev) * 256
t, w = heatmap.generate_target_batch(j, torch.ones_like(j), [256, 256], [64, 64], 2, True)
meta = {"target": t, "target_weight": w}
for _ in range(5): tr.step(img, meta)
torch.cuda.synchronize()
host = 0.0; t0 = time.perf_counter()
for _ in range(30):
    a = time.perf_counter(); tr.step(img, meta); host += time.perf_counter() - a
    torch.cuda.synchronize()          # isolate: host enqueue time without queueing back-pressure
tot_sync = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(30): tr.step(img, meta)
torch.cuda.synchronize()
print(f"{v}: host enqueue {host / 30 * 1e3:.2f} ms/step; step (async) {(time.perf_counter() - t0) / 30 * 1e3:.2f} ms; step (sync each) {tot_sync / 30 * 1e3:.2f} ms")

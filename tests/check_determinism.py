"""Two identical runs of two training steps per model (B, A, mynet, Lite-HRNet, hourglass; dropout on): prints whether outputs, loss,
parameter gradients and buffers agree bit for bit.  With LHN_DETERMINISTIC=1 they must (exit code 3 otherwise); without it
the script only reports (atomics make the last bits order-dependent).  Used by tests/test_train_gpu.py."""
import os, sys, hashlib
ROOT = os.environ.get("LHN_REPO") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # (tests/: may import the oracle's input generators)
sys.path.insert(0, ROOT)
import numpy as np, torch
from litehandnet_amd import _lib, get_loss, get_model
from litehandnet_amd.config import litehandnet_cfg
from oracle import synth
want = int(os.environ.get("LHN_DETERMINISTIC", "0") == "1")
assert _lib.lib().lhn_deterministic() == want
dev = torch.device("cuda:0")
out = []
for variant, size, n in (("B", 128, 8), ("A", 64, 4), ("M", 64, 4), ("L", 64, 4), ("H", 64, 2)):
    cfg = litehandnet_cfg(variant)
    crit = get_loss(cfg)
    x = synth.synth_images(n, size, 7).to(dev)
    t = torch.rand(n, 21, size // 4, size // 4, generator=torch.Generator().manual_seed(3)).to(dev)
    meta = {"target": t, "target_weight": torch.ones(n, 21, 1, device=dev)}
    digests = []
    for run in range(2):
        torch.manual_seed(11)                                   # same dropout masks in both runs
        m = get_model(cfg)
        m.load_state_dict({k: v.clone() for k, v in synth.synth_state_dict(m, 5).items()})
        m.to(dev).train()
        h = hashlib.sha256()
        for step in range(2):                                   # two steps: the second starts from moved running statistics
            y = m(x)
            loss, _ = crit(y, meta)
            m.zero_grad()
            loss.backward()
            h.update(y.detach().cpu().numpy().tobytes())
            h.update(np.float64(float(loss)).tobytes())
            for k, p in m.named_parameters():
                h.update(p.grad.detach().cpu().numpy().tobytes())
            for k, b in m.named_buffers():
                h.update(b.detach().cpu().numpy().tobytes())
        digests.append(h.hexdigest())
    out.append((variant, digests[0] == digests[1], digests[0][:12]))
print("DET", out, flush=True)
sys.exit(0 if (all(ok for _, ok, _ in out) or not want) else 3)

"""Debug aid: gradients of one model in the default mode against the same in LHN_DETERMINISTIC=1 (two child processes, since the
switch is read once per process).   python scripts/det_diff.py L 2 128"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(variant, n, size, out):
    import torch
    from litehandnet_amd import get_model
    from litehandnet_amd.config import litehandnet_cfg
    kw = dict(depth=18) if variant == "L" else {}
    cfg = litehandnet_cfg(variant, **kw)
    cfg.MODEL["ca_dropout"] = 0.0
    torch.manual_seed(5)
    m = get_model(cfg).cuda().train()
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(63)).standard_normal((n, 3, size, size)).astype(np.float32)).cuda()
    y = m(x)
    y = y[-1] if isinstance(y, (list, tuple)) else y
    g = torch.from_numpy(np.random.Generator(np.random.PCG64(164)).standard_normal(tuple(y.shape)).astype(np.float32)).cuda()
    y.backward(g)
    np.savez(out, y=y.detach().cpu().numpy(), **{k: p.grad.cpu().numpy() for k, p in m.named_parameters()})


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        sys.exit(0)
    v, n, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    outs = []
    for det in ("0", "1"):
        o = os.path.join(ROOT, "gpurun_out", f"det_{v}_{det}.npz")
        env = dict(os.environ, LHN_DETERMINISTIC=det)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", v, str(n), str(size), o], env=env, check=True)
        outs.append(np.load(o))
    a, b = outs
    gmax = max(float(np.linalg.norm(a[k])) for k in a.files if k != "y")
    print("forward", float(np.abs(a["y"] - b["y"]).max() / np.abs(a["y"]).max()))
    for k in a.files:
        if k == "y":
            continue
        e = float(np.linalg.norm(a[k].astype(np.float64) - b[k]) / (np.linalg.norm(a[k]) + 1e-3 * gmax))
        if e > 1e-4:
            print(f"{e:10.3e}  {k}  {a[k].shape}")

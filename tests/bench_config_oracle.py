"""CPU side of tests/test_zz_bench_config_gpu.py, run as a CHILD PROCESS (never touches the GPU): the float64 oracle (arbiter) and
the fp32 oracle (yardstick = what the reference computes on the CPU) of bench.py's workload -- batch 64, 256x256, train-mode
BatchNorm, Dropout2d p = 0.3 with seeded masks -- forward + TopdownHeatmapLoss + backward, for variants B and A.  It takes
1-2 minutes per variant on the host cores; conftest.py starts it at the beginning of a GPU session so that it runs UNDER the
other GPU tests instead of adding to the suite's wall time (round 2: 194 s of a 620 s suite).

    python tests/bench_config_oracle.py OUTDIR [B A]      ->  OUTDIR/bench_oracle_<variant>.npz (written atomically)
"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from litehandnet_amd import get_model  # noqa: E402
from litehandnet_amd.config import litehandnet_cfg  # noqa: E402
from litehandnet_amd.plan import PlanBuilder  # noqa: E402
from oracle import heatmap_np as onp  # noqa: E402
from oracle import synth, torch_ref  # noqa: E402

P, N, SIZE, SEED = 0.3, 64, 256, 7


def draw_masks(cfg, n, seed, p=P):
    """The masks tests/test_dropout_gpu.py::_attach draws: one [n, C] tensor per attention module in the PLAN's order (the order
    its launch records list them), generator PCG64([seed, index]).  The plan is built on the CPU (no library call)."""
    m = get_model(cfg)
    tensors = list(m.state_dict(keep_vars=True).values())
    pb = PlanBuilder(n, {id(t): j for j, t in enumerate(tensors)}, image_hw=(SIZE, SIZE), with_backward=True, p_drop=p)
    m.emit(pb, pb.image())
    names = {id(mod): k for k, mod in m.named_modules()}
    masks = {}
    for r in pb.recs:
        if r.get("mask") is not None:
            mod = r.get("ca", r.get("att"))
            k = names[id(mod)]
            if k not in masks:
                g = np.random.Generator(np.random.PCG64([seed, len(masks)]))
                masks[k] = torch.from_numpy(((g.random((n, r["y"].C)) < 1 - p) / (1 - p)).astype(np.float32))
    return masks


def run(variant, outdir):
    cfg = litehandnet_cfg(variant)
    ref = torch_ref.get_model(cfg, p_drop=P)
    ref.load_state_dict(synth.synth_state_dict(ref, SEED))
    ref.train()
    masks = draw_masks(cfg, N, SEED + 500)
    x = synth.synth_images(N, SIZE, SEED)
    j = synth.synth_joints(N, 21, SIZE, SEED + 1)
    tgt = torch.from_numpy(np.stack([onp.msra_generate_target(a, np.ones_like(a), [SIZE, SIZE], [64, 64])[0] for a in j]))
    tw = torch.ones(N, 21, 1)
    torch_ref.install_masks(ref, masks)
    ref32 = copy.deepcopy(ref)
    y32 = ref32(x)
    l32 = cfg.LOSS.loss_weight[0] * torch_ref.distance_loss(y32, tgt, tw)
    l32.backward()
    keys = [k for k, _ in ref32.named_parameters()]
    g32 = np.array([float(p.grad.norm()) for _, p in ref32.named_parameters()])
    y32 = y32.detach().numpy()
    del ref32
    ref = ref.double()
    y64 = ref(x.double())
    l64 = cfg.LOSS.loss_weight[0] * torch_ref.distance_loss(y64, tgt.double(), tw.double())
    l64.backward()
    g64 = np.array([float(p.grad.norm()) for _, p in ref.named_parameters()])
    tmp = os.path.join(outdir, f"bench_oracle_{variant}.tmp.npz")
    np.savez(tmp, y64=y64.detach().numpy(), y32=y32, l64=float(l64), l32=float(l32), keys=np.array(keys), g64=g64, g32=g32,
             mask_names=np.array(list(masks)), **{f"mask_{i}": v.numpy() for i, v in enumerate(masks.values())})
    os.replace(tmp, os.path.join(outdir, f"bench_oracle_{variant}.npz"))
    print("done", variant, flush=True)


if __name__ == "__main__":
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // 2)))
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    for v in (sys.argv[2:] or ["B", "A"]):
        run(v, out)

"""Round-3 fixtures.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

  init_weights.json     `init_weights` / `normal_init` of the REAL reference (liteHandNet.py:236-238, weight_init.py:28-32,
                        litehourglass.py:224-230, pose_hg_ms_att.py:256-262, hourglassnet.py, lite_hrnet.py) under
                        torch.manual_seed(SEED): sha256 over every tensor of `state_dict()` in order + a few per-key float64
                        sums.  The mirrors and the oracle, built under the same seed, must reproduce it bit for bit (row a15).
  scale_rotation.npz    TopDownGetRandomScaleRotation (datasets/data_pipeline/topdown_affine.py:11-45) of the REAL reference
                        under np.random.seed: scale / rotation per sample (the class never calls cv2; its module imports it).

    python tests/golden/make_golden_r3.py [init|scalerot|models|all]

  model_L30_128.npz, model_H1_256.npz   Lite-HRNet-30 and the 1-stack hourglass at 256 x 256 (see model_fixtures)
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from make_golden import _load_reference, by_path_ref  # noqa: E402
from litehandnet_amd import get_model  # noqa: E402
from litehandnet_amd.config import litehandnet_cfg  # noqa: E402
from oracle import torch_ref  # noqa: E402

SEED = 20261005
VARIANTS = (("A", {}), ("B", {}), ("M", {}), ("H", dict(num_stack=2)), ("L", dict(depth=18)), ("L", dict(depth=30)))


def state_digest(sd):
    h = hashlib.sha256()
    sums = {}
    for k, v in sd.items():
        a = v.detach().cpu().contiguous().numpy()
        h.update(k.encode())
        h.update(a.tobytes())
        sums[k] = float(a.astype(np.float64).sum())
    return h.hexdigest(), sums


def init_fixture(ref_models, ref_b):
    out = {"seed": SEED, "models": {}}
    for v, kw in VARIANTS:
        cfg = litehandnet_cfg(v, **kw)
        torch.manual_seed(SEED)
        r = ref_b.LiteHandNet(cfg) if v == "B" else ref_models.get_model(cfg)
        dig, sums = state_digest(r.state_dict())
        for build in (get_model, torch_ref.get_model):           # mirror and oracle reproduce the reference's initial state
            torch.manual_seed(SEED)
            d2, _ = state_digest(build(cfg).state_dict())
            assert d2 == dig, (v, kw, build.__module__)
        tag = v + "".join(f"_{k}{x}" for k, x in kw.items())
        gam = [x for k, x in r.state_dict().items() if x.dim() == 1 and k.endswith(".weight")]      # BatchNorm gammas
        out["models"][tag] = {"variant": v, "kw": kw, "class": type(r).__name__, "sha256": dig, "tensors": len(sums),
                              "params": sum(p.numel() for p in r.parameters()),
                              "bn_gamma_negative_fraction": float(np.mean([float((g < 0).float().mean()) for g in gam])) if gam else None,
                              "sums": {k: sums[k] for k in list(sums)[:6] + list(sums)[-3:]}}
        print(tag, type(r).__name__, dig[:16], out["models"][tag]["params"], out["models"][tag]["bn_gamma_negative_fraction"])
    json.dump(out, open(os.path.join(HERE, "init_weights.json"), "w"), indent=1, sort_keys=True)


def scalerot_fixture():
    for pkg in ("datasets", "datasets.data_pipeline"):
        sys.modules.setdefault(pkg, types.ModuleType(pkg))
    sys.modules["datasets.data_pipeline"].__path__ = []
    ta = by_path_ref("datasets.data_pipeline.topdown_affine", "datasets/data_pipeline/topdown_affine.py")
    aug = ta.TopDownGetRandomScaleRotation(rot_factor=40, scale_factor=0.5, rot_prob=0.6)
    np.random.seed(4242)
    n = 64
    scale0 = np.random.Generator(np.random.PCG64(7)).uniform(0.5, 1.5, (n, 2)).astype(np.float32)
    scales, rots = [], []
    for i in range(n):
        res = aug({"scale": scale0[i].copy()})
        scales.append(np.asarray(res["scale"], np.float64))
        rots.append(float(res["rotation"]))
    np.savez_compressed(os.path.join(HERE, "scale_rotation.npz"), seed=4242, scale_in=scale0, scale_out=np.stack(scales),
                        rotation=np.array(rots, np.float64), rot_factor=40, scale_factor=0.5, rot_prob=0.6)
    print("written scale_rotation.npz", float(np.mean(np.array(rots) == 0)))


def model_fixtures(ref_models, RefLoss):
    """The two config-5 sizes round 2 left without a reference fixture: Lite-HRNet-30 (config/litehrnet/_1_*_30.py) and the
    1-stack hourglass at 256 x 256 (config/hourglass/_3_*_h1.py).  Same recipe as make_golden.py::_model_case: the REAL reference
    and the oracle run forward + loss + backward on seeded inputs, must agree, and the reference's outputs are stored."""
    from make_golden import _model_case
    from make_golden_r2_models import _stacked
    from loss.heatmapLoss import DistanceLoss as RefDistance
    cfg = litehandnet_cfg("L", depth=30)
    r, o = ref_models.get_model(cfg), torch_ref.get_model(cfg)
    assert type(r).__name__ == "LiteHRNet" and sum(p.numel() for p in r.parameters()) == 1773361
    _model_case(r, o, 4, 128, 43, "L30_128", {"ref_loss": RefLoss(cfg), "ora_loss": torch_ref.TopdownHeatmapLoss(cfg)})
    cfg = litehandnet_cfg("H", num_stack=1)
    r, o = ref_models.get_model(cfg), torch_ref.get_model(cfg)
    assert sum(p.numel() for p in r.parameters()) == 3427733            # debug_litehandnet.ipynb:542
    lw = cfg.LOSS.loss_weight[0]
    _model_case(r, o, 2, 256, 34, "H1_256", {"ref_loss": _stacked(RefDistance(loss_type="L2", reduction="mean", balance=True), lw),
                                              "ora_loss": _stacked(torch_ref.distance_loss, lw)})


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ref_models, ref_b, RefLoss, pt, gt, ev = _load_reference()
    if what in ("models", "all"):
        torch.manual_seed(0)
        model_fixtures(ref_models, RefLoss)
    if what in ("init", "all"):
        init_fixture(ref_models, ref_b)
    if what in ("scalerot", "all"):
        scalerot_fixture()

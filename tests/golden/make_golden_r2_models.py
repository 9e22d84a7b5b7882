"""Round-2 model fixtures (imported by make_golden_r2.py; RUNS ONLY IN THE BUILD CONTAINER).

hourglass (models/pose_estimation/hourglassnet.py), num_stack 1 and 2: the REAL reference and the oracle restatement run
forward + loss + backward on seeded inputs with synthesised weights, must agree (forward 1e-5, gradient norms 2e-4, running
statistics), and the reference's outputs are stored.  Loss: the reference's DistanceLoss (loss/heatmapLoss.py:228-265) with
the target and weights repeated for every stack -- its TopdownHeatmapLoss hands a 4-D target to a 5-D output, which only
broadcasts when N == num_stack."""
import torch

from make_golden import _model_case
from litehandnet_amd.config import litehandnet_cfg
from oracle import torch_ref


def _stacked(lossfn, lw):
    def f(y, meta):
        S = y.shape[1]
        t = meta["target"].unsqueeze(1).expand(-1, S, -1, -1, -1).contiguous()
        w = meta["target_weight"].unsqueeze(1).expand(-1, S, -1, -1).contiguous()
        return lw * lossfn(y, t, w), None
    return f


def hourglass_fixtures(ref_models, RefLoss):
    from loss.heatmapLoss import DistanceLoss as RefDistance
    known = {1: 3427733}                                  # debug_litehandnet.ipynb:542 (1 stack, C = 256)
    for ns, n, size, seed in ((1, 4, 128, 31), (2, 2, 128, 32), (2, 2, 256, 33)):
        cfg = litehandnet_cfg("H", num_stack=ns)
        r, o = ref_models.get_model(cfg), torch_ref.get_model(cfg)
        assert type(r).__name__ == "HourglassNet" and list(r.state_dict()) == list(o.state_dict())
        npar = sum(p.numel() for p in r.parameters())
        assert ns not in known or npar == known[ns], npar
        lw = cfg.LOSS.loss_weight[0]
        out = {"ref_loss": _stacked(RefDistance(loss_type="L2", reduction="mean", balance=True), lw),
               "ora_loss": _stacked(torch_ref.distance_loss, lw)}
        _model_case(r, o, n, size, seed, f"H{ns}_{size}", out)
        print(f"hourglass num_stack={ns}: {npar} parameters")


def litehrnet_fixtures(ref_models, RefLoss):
    """Lite-HRNet (models/pose_estimation/lite_hrnet.py), depth 18: the REAL reference vs the oracle restatement (forward,
    TopdownHeatmapLoss, gradient norms, running statistics -- including the fuse layers the reference evaluates twice per
    forward), known answer 1,483,873 parameters (test_models_performance.ipynb:276-279)."""
    cfg = litehandnet_cfg("L", depth=18)
    out = {"ref_loss": RefLoss(cfg), "ora_loss": torch_ref.TopdownHeatmapLoss(cfg)}
    for n, size, seed in ((4, 128, 41), (2, 256, 42)):
        r, o = ref_models.get_model(cfg), torch_ref.get_model(cfg)
        assert type(r).__name__ == "LiteHRNet" and list(r.state_dict()) == list(o.state_dict())
        assert sum(p.numel() for p in r.parameters()) == 1483873
        _model_case(r, o, n, size, seed, f"L18_{size}", out)

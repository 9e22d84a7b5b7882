"""Fixtures for the options added late in round 1.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

  model_Asilu_64.npz  variant A with cfg.MODEL.activation = 'silu' (liteHandNet.py:203-205 -> nn.SiLU everywhere)
  model_Mact_128.npz  mynet with cfg.MODEL.output_acitivation = True (pose_hg_ms_att.py:232,251-252, the reference's spelling)
  decode_udp.npz      transform_preds(..., use_udp=True) (post_transforms.py:6-48, the (W-1) scaling of the UDP configs)

Same recipe as make_golden.py::_model_case: the REAL reference and the oracle run forward + TopdownHeatmapLoss + backward on
seeded inputs with synthesised weights, must agree, and the reference's outputs are stored.

    python tests/golden/make_golden_extra.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from make_golden import _load_reference, _model_case  # noqa: E402
from litehandnet_amd.config import litehandnet_cfg  # noqa: E402
from oracle import torch_ref  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, _, RefLoss, pt, _, _ = _load_reference()
    cfgA = litehandnet_cfg("A")
    out = {"ref_loss": RefLoss(cfgA), "ora_loss": torch_ref.TopdownHeatmapLoss(cfgA)}

    cfgS = litehandnet_cfg("A", activation="silu")
    rS, oS = ref_models.get_model(cfgS), torch_ref.get_model(cfgS)
    assert list(rS.state_dict()) == list(oS.state_dict())
    assert any(isinstance(m, torch.nn.SiLU) for m in rS.modules())
    _model_case(rS, oS, 8, 64, 21, "Asilu_64", out)

    cfgM = litehandnet_cfg("M", output_acitivation=True)
    rM, oM = ref_models.get_model(cfgM), torch_ref.get_model(cfgM)
    assert rM.with_activation and list(rM.state_dict()) == list(oM.state_dict())
    _model_case(rM, oM, 8, 128, 22, "Mact_128", out)
    import numpy as np
    from oracle import heatmap_np as onp
    r = np.random.Generator(np.random.PCG64(41))
    coords = r.uniform(-2, 66, (8, 21, 2)).astype(np.float32)
    center = r.uniform(60, 200, (8, 2)).astype(np.float32)
    scale = r.uniform(0.5, 1.5, (8, 2)).astype(np.float32)
    res = {}
    for udp in (False, True):
        a = np.stack([pt.transform_preds(coords[i].copy(), center[i], scale[i], [64, 48], use_udp=udp) for i in range(8)])
        b = np.stack([onp.transform_preds(coords[i].copy(), center[i], scale[i], [64, 48], use_udp=udp) for i in range(8)])
        assert np.array_equal(a, b), udp
        res["udp" if udp else "plain"] = a
    np.savez_compressed(os.path.join(HERE, "decode_udp.npz"), coords=coords, center=center, scale=scale, output_size=np.array([64, 48]), **res)
    print("written model_Asilu_64.npz, model_Mact_128.npz, decode_udp.npz")


if __name__ == "__main__":
    main()

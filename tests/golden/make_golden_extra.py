"""Fixtures for the options added late in round 1.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

  model_Asilu_64.npz  variant A with cfg.MODEL.activation = 'silu' (liteHandNet.py:203-205 -> nn.SiLU everywhere)
  model_Mact_128.npz  mynet with cfg.MODEL.output_acitivation = True (pose_hg_ms_att.py:232,251-252, the reference's spelling)

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
    ref_models, _, RefLoss, _, _, _ = _load_reference()
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
    print("written model_Asilu_64.npz, model_Mact_128.npz")


if __name__ == "__main__":
    main()

"""Fixtures for the options added late in round 1.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

  model_Asilu_64.npz  variant A with cfg.MODEL.activation = 'silu' (liteHandNet.py:203-205 -> nn.SiLU everywhere)
  model_Mact_128.npz  mynet with cfg.MODEL.output_acitivation = True (pose_hg_ms_att.py:232,251-252, the reference's spelling)
  decode_udp.npz      transform_preds(..., use_udp=True) (post_transforms.py:6-48, the (W-1) scaling of the UDP configs)
  decode_legacy.npz   adjust_keypoints_by_offset (utils/heatmap_post_processing.py:6-33) and the 11x11 peak NMS
                      (utils/result_parser.py:50-59 = torch max_pool2d + eq + mul)

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
    # legacy +-0.25 offset refinement (clamped neighbours, then +0.5) and the heat-map peak NMS, straight from the
    # reference's files (heatmap_post_processing.py imports cv2 and config.pcfg at module level; neither is used here)
    import types
    from make_golden import by_path_ref
    if "config" not in sys.modules:                      # config/__init__.py needs `addict` (absent); only the unused `pcfg` name is imported
        stub = types.ModuleType("config")
        stub.pcfg = {}
        sys.modules["config"] = stub
    hp = by_path_ref("ref_heatmap_post_processing", "utils/heatmap_post_processing.py")
    hm = r.random((3, 21, 24, 20)).astype(np.float32)
    hm[0, 0, 0, 0] = 2.0          # peaks on the border: the clamped-neighbour branches
    hm[0, 1, 23, 19] = 2.0
    hm[0, 2, 0, 19] = 2.0
    p0, mv = onp.get_max_preds(hm)
    kp = torch.from_numpy(np.concatenate([p0, mv], 2).copy())
    ref_adj = hp.adjust_keypoints_by_offset(kp.clone(), torch.from_numpy(hm)).numpy()[..., :2]
    assert np.array_equal(ref_adj, onp.refine_offset_legacy(hm, p0))
    t = torch.from_numpy(hm.copy())
    mx = torch.nn.functional.max_pool2d(t, 11, 1, 5)                 # result_parser.py:25-27,50-59 with nms_kernel 11
    ref_nms = (t * torch.eq(mx, t).float()).numpy()
    assert np.array_equal(ref_nms, onp.heatmap_nms(hm, 11))
    np.savez_compressed(os.path.join(HERE, "decode_legacy.npz"), heatmaps=hm, argmax_xy=p0, adjusted=ref_adj,
                        nms_nonzero=np.argwhere(ref_nms != 0).astype(np.int32), nms_sum=np.float64(ref_nms.astype(np.float64).sum()))
    print("written model_Asilu_64.npz, model_Mact_128.npz, decode_udp.npz, decode_legacy.npz")


if __name__ == "__main__":
    main()

"""Fixtures for the options added late in round 1.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

  model_Asilu_64.npz  variant A with cfg.MODEL.activation = 'silu' (liteHandNet.py:203-205 -> nn.SiLU everywhere)
  model_Mact_128.npz  mynet with cfg.MODEL.output_acitivation = True (pose_hg_ms_att.py:232,251-252, the reference's spelling)
  decode_udp.npz      transform_preds(..., use_udp=True) (post_transforms.py:6-48, the (W-1) scaling of the UDP configs)
  affine.npz          the geometry of TopDownAffine (post_transforms.py:52-156): get_warp_matrix / warp_affine_joints outputs,
                      and the three source / destination points get_affine_transform hands to cv2.getAffineTransform
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
    # TopDownAffine geometry.  get_affine_transform ends in cv2.getAffineTransform (absent), so its three point pairs are
    # rebuilt here with the reference's own helpers (post_transforms.py:129-147) and the oracle's closed form has to map
    # one triplet onto the other; get_warp_matrix / warp_affine_joints are pure numpy and are compared directly.
    ra = np.random.Generator(np.random.PCG64(43))     # own stream: the sections below keep theirs
    cs = ra.uniform(40, 220, (12, 2)).astype(np.float32)
    ss = ra.uniform(0.4, 1.6, (12, 1)).astype(np.float32).repeat(2, 1)
    rots = ra.uniform(-60, 60, 12).astype(np.float32)
    rots[0] = 0.0
    osz = np.array([256, 192], np.float32)
    src_pts, dst_pts, udp_m, udp_j = [], [], [], []
    jj = ra.uniform(0, 255, (12, 21, 2)).astype(np.float32)
    for c, sc, ro, jn in zip(cs, ss, rots, jj):
        scale_tmp = sc * 200.0
        src_dir = pt.rotate_point([0., scale_tmp[0] * -0.5], np.pi * ro / 180)
        dst_dir = np.array([0., osz[0] * -0.5])
        src = np.zeros((3, 2), np.float32)
        src[0, :] = c
        src[1, :] = c + src_dir
        src[2, :] = pt._get_3rd_point(src[0, :], src[1, :])
        dst = np.zeros((3, 2), np.float32)
        dst[0, :] = [osz[0] * 0.5, osz[1] * 0.5]
        dst[1, :] = np.array([osz[0] * 0.5, osz[1] * 0.5]) + dst_dir
        dst[2, :] = pt._get_3rd_point(dst[0, :], dst[1, :])
        M = onp.affine_matrix(c, sc, ro, osz)
        got = src.astype(np.float64) @ M[:, :2].T + M[:, 2]
        assert np.abs(got - dst).max() < 2e-4, np.abs(got - dst).max()
        src_pts.append(src)
        dst_pts.append(dst)
        mu = pt.get_warp_matrix(ro, c * 2.0, osz - 1.0, sc * 200.0)          # topdown_affine.py:76-93
        assert np.allclose(mu, onp.warp_matrix_udp(ro, c * 2.0, osz - 1.0, sc * 200.0), rtol=1e-5, atol=1e-4)
        udp_m.append(mu)
        udp_j.append(pt.warp_affine_joints(jn, mu))
    np.savez_compressed(os.path.join(HERE, "affine.npz"), center=cs, scale=ss, rot=rots, output_size=osz, src=np.stack(src_pts),
                        dst=np.stack(dst_pts), joints=jj, udp_matrix=np.stack(udp_m), udp_joints=np.stack(udp_j))

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
    print("written model_Asilu_64.npz, model_Mact_128.npz, decode_udp.npz, affine.npz, decode_legacy.npz")


if __name__ == "__main__":
    main()

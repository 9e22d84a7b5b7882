"""Round-2 fixtures.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

  decoder_result.npz   TopDownDecoder.decode / decode_simdr of the REAL reference (utils/post_processing/decoder.py:26-107)
                       on seeded heat maps / SimDR vectors and meta: every key of the result dict ('default' post-process; the
                       'unbiased' one needs cv2.GaussianBlur, absent here)
  model_H{1,2}_*.npz   hourglass (models/pose_estimation/hourglassnet.py), num_stack 1 and 2: forward [N,S,K,H,W], loss,
                       gradient norms -- same recipe as make_golden.py::_model_case

    python tests/golden/make_golden_r2.py [decoder|hourglass|all]
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from make_golden import _load_reference, by_path_ref  # noqa: E402
from litehandnet_amd.config import litehandnet_cfg  # noqa: E402
from oracle import heatmap_np as onp  # noqa: E402
from oracle import synth, torch_ref  # noqa: E402


def decoder_fixture(ev):
    # decoder.py imports keypoints_from_heatmaps / keypoints_from_simdr through the package path; hand it the module that
    # make_golden loaded by file path (the package __init__ chain needs mmcv / torchvision)
    for pkg in ("utils", "utils.post_processing", "utils.post_processing.evaluation"):
        sys.modules.setdefault(pkg, types.ModuleType(pkg))
    sys.modules["utils.post_processing.evaluation.top_down_eval"] = ev
    dec = by_path_ref("ref_decoder", "utils/post_processing/decoder.py")
    cfg = litehandnet_cfg("B")
    cfg.PIPELINE["unbiased_encoding"] = False
    cfg.PIPELINE["simdr_split_ratio"] = 2
    d = dec.TopDownDecoder(cfg)
    assert d.post_process == "default" and d.k == 2
    r = np.random.Generator(np.random.PCG64(91))
    n = 6
    hm = r.random((n, 24, 64, 64)).astype(np.float32)          # 3 extra channels: decode() keeps the first num_joints
    hm[0, 0] = 0.0                                            # an all-zero map: coordinates -1 (maxval <= 0)
    hm[1, 2, 0, 5] = 3.0                                      # a border peak: no +-0.25 shift
    meta = dict(bbox_score=torch.from_numpy(r.random(n).astype(np.float32)), bbox_id=torch.arange(100, 100 + n),
                image_file=[f"img_{i}.jpg" for i in range(n)],
                center=torch.from_numpy(r.uniform(60, 200, (n, 2)).astype(np.float32)),
                scale=torch.from_numpy(r.uniform(0.5, 1.5, (n, 2)).astype(np.float32)),
                simdr_x=torch.from_numpy(r.random((n, 21, 512)).astype(np.float32)),
                simdr_y=torch.from_numpy(r.random((n, 21, 512)).astype(np.float32)))
    res = d.decode(meta, torch.from_numpy(hm))
    rs = d.decode_simdr(meta, torch.from_numpy(hm))
    assert sorted(res) == ["bbox_ids", "boxes", "hm_preds", "image_paths", "output_heatmap", "preds"]
    assert sorted(rs) == ["bbox_ids", "boxes", "image_paths", "output_heatmap", "preds"]
    np.savez_compressed(os.path.join(HERE, "decoder_result.npz"), heatmaps=hm, bbox_score=meta["bbox_score"].numpy(),
                        bbox_id=meta["bbox_id"].numpy(), center=meta["center"].numpy(), scale=meta["scale"].numpy(),
                        simdr_x=meta["simdr_x"].numpy(), simdr_y=meta["simdr_y"].numpy(),
                        preds=res["preds"], hm_preds=res["hm_preds"], boxes=res["boxes"], bbox_ids=np.array(res["bbox_ids"]),
                        out_heatmap_sum=np.float64(res["output_heatmap"].astype(np.float64).sum()),
                        out_heatmap_shape=np.array(res["output_heatmap"].shape),
                        simdr_preds=rs["preds"], simdr_boxes=rs["boxes"],
                        dtypes=np.array([str(res[k].dtype) for k in ("preds", "hm_preds", "boxes", "output_heatmap")] + [str(rs["preds"].dtype)]))
    print("written decoder_result.npz", {k: getattr(v, "shape", None) for k, v in res.items()}, rs["preds"].dtype)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, _, RefLoss, pt, _, ev = _load_reference()
    if what in ("decoder", "all"):
        decoder_fixture(ev)
    if what in ("hourglass", "all"):
        from make_golden_r2_models import hourglass_fixtures
        hourglass_fixtures(ref_models, RefLoss)


if __name__ == "__main__":
    main()

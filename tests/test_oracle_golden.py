"""CPU: the oracle restatement against the committed golden vectors (generated from the real reference
by tests/golden/make_golden.py).  Nothing here touches /root/reference."""
import json
import os

import numpy as np
import torch

from litehandnet_amd.config import litehandnet_cfg
from oracle import heatmap_np as onp
from oracle import synth, torch_ref


def _no_dropout(m):
    for x in m.modules():
        if isinstance(x, (torch.nn.Dropout2d, torch.nn.Dropout)):
            x.p = 0.0


def test_known_answers(golden_dir):
    ka = json.load(open(os.path.join(golden_dir, "known_answers.json")))
    a = torch_ref.get_model(litehandnet_cfg("A"))
    b = torch_ref.get_model(litehandnet_cfg("B"))
    # 2,272,981 is the reference's own published count (test_models_performance.ipynb:247)
    assert sum(p.numel() for p in a.parameters()) == ka["litehandnet_A_params"] == 2272981
    assert sum(p.numel() for p in b.parameters()) == ka["litehourglass_B_params"]
    assert len(a.state_dict()) == ka["A_keys"] and len(b.state_dict()) == ka["B_keys"]


def _run_case(golden_dir, tag, variant, **kw):
    g = np.load(os.path.join(golden_dir, f"model_{tag}.npz"))
    cfg = litehandnet_cfg(variant, **kw)
    m = torch_ref.get_model(cfg)
    n, size, seed = int(g["n"]), int(g["size"]), int(g["seed"])
    m.load_state_dict(synth.synth_state_dict(m, seed))
    m.train()
    _no_dropout(m)
    x = synth.synth_images(n, size, seed)
    y = m(x)
    assert np.abs(y.detach().numpy() - g["heatmap"]).max() <= 1e-5 * np.abs(g["heatmap"]).max()
    hs = size // 4
    j = synth.synth_joints(n, 21, size, seed + 1)
    tgt = np.stack([onp.msra_generate_target(a, np.ones_like(a), [size, size], [hs, hs])[0] for a in j])
    meta = {"target": torch.from_numpy(tgt), "target_weight": torch.from_numpy(g["target_weight"])}
    loss, _ = torch_ref.TopdownHeatmapLoss(cfg)(y, meta)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    loss.backward()
    gn = dict(zip(g["grad_keys"].tolist(), g["grad_norms"].tolist()))
    for k, p in m.named_parameters():
        assert abs(float(p.grad.norm()) - gn[k]) <= 2e-4 * (gn[k] + 1e-6), k
    bk = str(g["bn_key"])
    assert np.allclose(m.state_dict()[bk].numpy(), g["bn_running_mean"], rtol=1e-5, atol=1e-6)


def test_model_B_64(golden_dir):
    _run_case(golden_dir, "B_64", "B")


def test_model_A_64(golden_dir):
    _run_case(golden_dir, "A_64", "A")


def test_model_Bca_64(golden_dir):
    _run_case(golden_dir, "Bca_64", "B", rbu_ca="ca")


def test_model_M_128(golden_dir):
    """`mynet` (models/pose_hg_ms_att.py): oracle vs the reference-generated fixture; 2,240,405 parameters."""
    _run_case(golden_dir, "M_128", "M")
    m = torch_ref.get_model(litehandnet_cfg("M"))
    assert sum(p.numel() for p in m.parameters()) == 2240405
    g = np.load(os.path.join(golden_dir, "model_M_64_eval.npz"))
    m.load_state_dict(synth.synth_state_dict(m, int(g["weights_seed"])))
    m.eval()
    with torch.no_grad():
        y = m(synth.synth_images(2, 64, int(g["seed"]))).numpy()
    assert np.abs(y - g["heatmap"]).max() <= 1e-5 * np.abs(g["heatmap"]).max()


def test_model_Bse_64(golden_dir):
    _run_case(golden_dir, "Bse_64", "B", msrb_ca="se", rbu_ca="se")


def test_eval_mode(golden_dir):
    for tag, variant, wseed in (("B", "B", 5), ("A", "A", 6)):
        g = np.load(os.path.join(golden_dir, f"model_{tag}_64_eval.npz"))
        m = torch_ref.get_model(litehandnet_cfg(variant))
        m.load_state_dict(synth.synth_state_dict(m, wseed))
        m.eval()
        with torch.no_grad():
            y = m(synth.synth_images(2, 64, int(g["seed"]))).numpy()
        assert np.abs(y - g["heatmap"]).max() <= 1e-5 * np.abs(g["heatmap"]).max()


def test_deploy_reparam(golden_dir):
    """Oracle deploy_model() vs the fixture the real reference produced (keys, parameter count, fused tensors, output).
    2-ulp tolerance on the tensors: torch's CPU sqrt differs between hosts (MKL VML)."""
    for variant in ("B", "A"):
        g = np.load(os.path.join(golden_dir, f"model_{variant}_64_deploy.npz"))
        m = torch_ref.get_model(litehandnet_cfg(variant))
        m.load_state_dict(synth.synth_state_dict(m, int(g["weights_seed"])))
        m.eval()
        torch_ref.deploy_model(m)
        sd = m.state_dict()
        assert list(sd) == [str(k) for k in g["keys"]]
        assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
        for k in g.files:
            if k.startswith("t_"):
                assert np.all(np.abs(sd[k[2:]].numpy() - g[k]) <= 2.4e-7 * np.abs(g[k]) + 3e-7 * np.abs(g[k]).max()), k
        with torch.no_grad():
            y = m(synth.synth_images(2, 64, int(g["seed"]))).numpy()
        assert np.abs(y - g["heatmap"]).max() <= 1e-5 * np.abs(g["heatmap"]).max()
    assert int(np.load(os.path.join(golden_dir, "model_A_64_deploy.npz"))["n_params"]) == 2265621   # notebook known answer


def test_encode(golden_dir):
    g = np.load(os.path.join(golden_dir, "encode.npz"))
    for unb, tag in ((True, "unbiased"), (False, "biased")):
        for i, (j, v) in enumerate(zip(g["joints"], g["visible"])):
            t, w = onp.msra_generate_target(j, v, [256, 256], [64, 64], 2, unb)
            flat = t.reshape(21, -1)
            assert np.array_equal(w, g[f"{tag}_weight"][i])
            assert np.array_equal(flat.argmax(1), g[f"{tag}_argmax"][i])
            assert np.array_equal(flat.max(1), g[f"{tag}_max"][i])
            if i < 2:
                assert np.array_equal(t, g[f"{tag}_full_first2"][i])


def test_decode_and_metrics(golden_dir):
    g = np.load(os.path.join(golden_dir, "decode.npz"))
    p0, mv = onp.get_max_preds(g["heatmaps"])
    assert np.array_equal(p0, g["argmax_xy"]) and np.array_equal(mv, g["maxvals"])
    hp, pr, _ = onp.keypoints_from_heatmaps(g["heatmaps"], g["center"], g["scale"], "default")
    assert np.array_equal(hp, g["hm_preds"]) and np.array_equal(pr, g["preds"])
    acc, pck, cnt = onp.keypoint_pck_accuracy(g["preds"].copy(), g["gt"].copy(), g["mask"].copy(), 0.2, g["normalize"].copy())
    assert np.array_equal(acc, g["pck_acc"]) and pck == float(g["pck"]) and cnt == int(g["pck_cnt"])
    assert abs(onp.keypoint_auc(g["preds"].copy(), g["gt"].copy(), g["mask"].copy(), 30) - float(g["auc"])) < 1e-12
    assert abs(onp.keypoint_epe(g["preds"].copy(), g["gt"].copy(), g["mask"].copy()) - float(g["epe"])) < 1e-6
    p0i = g["argmax_xy"][1]
    tay = np.stack([onp.taylor(g["taylor_in"][k], p0i[k].copy()) for k in range(21)])
    assert np.allclose(tay, g["taylor_out"], rtol=1e-6, atol=1e-6)


def test_loss(golden_dir):
    g = np.load(os.path.join(golden_dir, "loss.npz"))
    r = np.random.Generator(np.random.PCG64(int(g["seed"])))
    o = torch.from_numpy(r.standard_normal((2, 21, 64, 64)).astype(np.float32)).requires_grad_()
    tw = [onp.msra_generate_target(a, v, [256, 256], [64, 64]) for a, v in zip(g["joints"], g["visible"])]
    t = torch.from_numpy(np.stack([a for a, _ in tw]))
    w = torch.from_numpy(np.stack([b for _, b in tw]))
    assert int((t > 0.5).sum()) == int(g["npos"])
    l = torch_ref.distance_loss(o, t, w)
    assert abs(float(l) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    l.backward()
    assert np.allclose(o.grad.numpy()[:, ::5, ::16, ::16], g["grad_sample"], rtol=1e-5, atol=1e-9)


def test_simdr(golden_dir):
    """SimDR targets / loss / decode of the oracle against the reference-generated fixture."""
    g = np.load(os.path.join(golden_dir, "simdr.npz"))
    js, vs = g["joints"], g["visible"]
    sx, sy = zip(*[onp.generate_sa_simdr(a, v, [256, 256], 2, 2) for a, v in zip(js, vs)])
    sx, sy = np.stack(sx), np.stack(sy)
    assert np.allclose(sx.sum(2), g["tx_sum"], rtol=1e-6) and np.array_equal(sx.argmax(2), g["tx_argmax"])
    cfg = litehandnet_cfg("B")
    cfg.PIPELINE["simdr_split_ratio"] = 2
    m = torch_ref.SimDRLoss(cfg)
    m.load_state_dict(synth.synth_state_dict(m, int(g["weights_seed"])))
    hm = torch.from_numpy(np.random.Generator(np.random.PCG64(int(g["hm_seed"]))).standard_normal((4, 21, 64, 64)).astype(np.float32) * 0.1)
    hm.requires_grad_()
    l = m(hm, torch.from_numpy(sx), torch.from_numpy(sy), torch.from_numpy(vs[..., :1].copy()))
    l.backward()
    assert abs(float(l) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert abs(float(hm.grad.abs().sum()) - float(g["dheatmap_abs_sum"])) <= 1e-4 * float(g["dheatmap_abs_sum"])
    assert np.array_equal(onp.keypoints_from_simdr(sx, sy, g["center"], g["scale"], 2), g["keypoints"])


def test_encode_udp(golden_dir):
    g = np.load(os.path.join(golden_dir, "encode.npz"))
    for i, (j, v) in enumerate(zip(g["joints"], g["visible"])):
        t, w = onp.udp_generate_target(j, v, [256, 256], [64, 64], 2)
        flat = t.reshape(21, -1)
        assert np.array_equal(w, g["udp_weight"][i]) and np.array_equal(flat.argmax(1), g["udp_argmax"][i])
        assert np.array_equal(flat.max(1), g["udp_max"][i])
        if i < 2:
            assert np.array_equal(t, g["udp_full_first2"][i])


def test_training_trajectory(golden_dir):
    """The oracle under torch Adam follows the REAL reference's 6-step trajectory (tests/golden/make_golden_train.py).
    Step 1 is tight; later steps are ill-conditioned (Adam moves noise-gradient parameters by +-lr), so the bar there is
    the reference's own fp32-vs-float64 drift stored in the fixture."""
    g = np.load(os.path.join(golden_dir, "train_B_64.npz"))
    n, size = int(g["n"]), int(g["size"])
    cfg = litehandnet_cfg("B")
    m = torch_ref.get_model(cfg, p_drop=0.0)
    m.load_state_dict(synth.synth_state_dict(m, int(g["weights_seed"])))
    m.train()
    crit = torch_ref.TopdownHeatmapLoss(cfg)
    opt = torch.optim.Adam(m.parameters(), lr=float(g["lr"]))
    losses = []
    for s in range(int(g["steps"])):
        x = synth.synth_images(n, size, 50 + 2 * s)
        j = synth.synth_joints(n, 21, size, 51 + 2 * s)
        t = np.stack([onp.msra_generate_target(a, np.ones_like(a), [size, size], [size // 4, size // 4])[0] for a in j])
        loss, _ = crit(m(x), {"target": torch.from_numpy(t), "target_weight": torch.ones(n, 21, 1)})
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    ref, f64 = g["losses"], g["losses_f64"]
    assert abs(losses[0] - ref[0]) <= 1e-6 * abs(ref[0])
    for s in range(1, len(ref)):
        assert abs(losses[s] - ref[s]) <= max(5 * abs(ref[s] - f64[s]), 2e-2 * abs(ref[s])), (s, losses, ref.tolist())


def test_model_A_silu_64(golden_dir):
    """cfg.MODEL.activation='silu' (row a1): fixture from the real reference (tests/golden/make_golden_extra.py)."""
    _run_case(golden_dir, "Asilu_64", "A", activation="silu")


def test_model_M_output_activation_128(golden_dir):
    """mynet with output_acitivation=True: leaky_relu(preds, 0.5) on the head (pose_hg_ms_att.py:251-252)."""
    _run_case(golden_dir, "Mact_128", "M", output_acitivation=True)


def test_transform_preds_udp(golden_dir):
    """post_transforms.py:6-48 with and without use_udp, non-square heatmap: oracle vs the reference's vectors."""
    g = np.load(os.path.join(golden_dir, "decode_udp.npz"))
    for key, udp in (("plain", False), ("udp", True)):
        out = np.stack([onp.transform_preds(g["coords"][i].copy(), g["center"][i], g["scale"][i], g["output_size"].tolist(), use_udp=udp)
                        for i in range(len(g["coords"]))])
        assert np.array_equal(out, g[key]), key


def test_legacy_offset_and_nms(golden_dir):
    """utils/heatmap_post_processing.py:6-33 (clamped +-0.25 offset, then +0.5) and the 11x11 peak NMS
    (utils/result_parser.py:50-59) on a non-square map with border peaks: oracle vs the reference's vectors."""
    g = np.load(os.path.join(golden_dir, "decode_legacy.npz"))
    hm = g["heatmaps"]
    p0, _ = onp.get_max_preds(hm)
    assert np.array_equal(p0, g["argmax_xy"])
    assert np.array_equal(onp.refine_offset_legacy(hm, p0), g["adjusted"])
    nms = onp.heatmap_nms(hm, 11)
    assert np.array_equal(np.argwhere(nms != 0).astype(np.int32), g["nms_nonzero"])
    assert nms.astype(np.float64).sum() == float(g["nms_sum"])


def test_affine_geometry(golden_dir):
    """TopDownAffine geometry (post_transforms.py:52-156).  The closed-form matrix of the oracle maps the three source points
    the reference builds for cv2.getAffineTransform onto its three destination points; the UDP matrix and the joint mapping
    (pure numpy in the reference) are compared directly.  Only cv2.warpAffine's interpolation stays unpinned."""
    g = np.load(os.path.join(golden_dir, "affine.npz"))
    osz = g["output_size"]
    for i in range(len(g["center"])):
        M = onp.affine_matrix(g["center"][i], g["scale"][i], g["rot"][i], osz)
        got = g["src"][i].astype(np.float64) @ M[:, :2].T + M[:, 2]
        assert np.abs(got - g["dst"][i]).max() < 2e-4
        Mu = onp.warp_matrix_udp(g["rot"][i], g["center"][i] * 2.0, osz - 1.0, g["scale"][i] * 200.0)
        assert np.allclose(Mu, g["udp_matrix"][i], rtol=1e-5, atol=1e-4)
        j = g["joints"][i].astype(np.float64)
        mapped = np.concatenate([j, np.ones((len(j), 1))], 1) @ Mu.T
        assert np.abs(mapped - g["udp_joints"][i]).max() < 1e-3


def test_dark_round_trip_property():
    """Size-independent property of the unpinned part (DARK needs cv2.GaussianBlur, absent): encoding a sub-pixel joint with the
    unbiased Gaussian and decoding it with blur + log + Taylor step recovers the joint to a few hundredths of a heat-map
    pixel, for both blur kernels the configs use (11 and 19); the plain argmax alone is off by up to half a pixel."""
    r = np.random.Generator(np.random.PCG64(77))
    j = np.zeros((4, 21, 3), np.float32)
    j[..., :2] = r.uniform(24, 232, (4, 21, 2))                     # image pixels, well inside a 256x256 crop
    hm = np.stack([onp.msra_generate_target(a, np.ones_like(a), [256, 256], [64, 64], 2, True)[0] for a in j])
    center = np.full((4, 2), 128.0, np.float32)
    scale = np.full((4, 2), 256.0 / 200.0, np.float32)             # identity back-transform: heat-map px * 4
    plain, _, _ = onp.keypoints_from_heatmaps(hm, center, scale, None)
    assert np.abs(plain - j[..., :2] / 4).max() <= 0.5 + 1e-6
    for k in (11, 19):
        hp, pr, _ = onp.keypoints_from_heatmaps(hm, center, scale, "unbiased", k)
        assert np.abs(hp - j[..., :2] / 4).max() < 0.05, k
        assert np.abs(pr - j[..., :2]).max() < 0.2, k


def test_decoder_result_fixture(golden_dir):
    """The oracle's decode restatement against the REAL reference's TopDownDecoder result dicts (make_golden_r2.py;
    utils/post_processing/decoder.py:26-107), 'default' post-process and SimDR: bit-exact."""
    g = np.load(os.path.join(golden_dir, "decoder_result.npz"))
    hm = g["heatmaps"][:, :21]
    hp, pr, mv = onp.keypoints_from_heatmaps(hm, g["center"], g["scale"], "default", 11)
    assert np.array_equal(np.concatenate([pr, mv], 2), g["preds"])
    assert np.array_equal(np.concatenate([hp * 4, mv], 2), g["hm_preds"])
    boxes = np.zeros((hm.shape[0], 6), np.float32)
    boxes[:, 0:2], boxes[:, 2:4] = g["center"], g["scale"]
    boxes[:, 4] = np.prod(g["scale"] * 200.0, axis=1)
    boxes[:, 5] = g["bbox_score"]
    assert np.array_equal(boxes, g["boxes"]) and np.array_equal(boxes, g["simdr_boxes"])
    sp = onp.keypoints_from_simdr(g["simdr_x"], g["simdr_y"], g["center"], g["scale"], 2)
    assert np.array_equal(sp.astype(np.float32), g["simdr_preds"])
    assert g["bbox_ids"].tolist() == g["bbox_id"].tolist()


def test_hourglass_fixtures(golden_dir):
    """oracle.torch_ref.HourglassNet against the REAL reference's vectors (make_golden_r2_models.py): forward heat maps
    [N, S, K, H, W] bit-for-bit, known parameter count 3,427,733 (debug_litehandnet.ipynb:542)."""
    for tag, ns in (("H1_128", 1), ("H2_128", 2)):
        g = np.load(os.path.join(golden_dir, f"model_{tag}.npz"))
        cfg = litehandnet_cfg("H", num_stack=ns)
        m = torch_ref.get_model(cfg)
        if ns == 1:
            assert sum(p.numel() for p in m.parameters()) == 3427733
        m.load_state_dict(synth.synth_state_dict(m, int(g["seed"])))
        m.train()
        y = m(synth.synth_images(int(g["n"]), int(g["size"]), int(g["seed"])))
        assert tuple(y.shape) == g["heatmap"].shape and y.shape[1] == ns
        assert np.abs(y.detach().numpy() - g["heatmap"]).max() <= 1e-5 * np.abs(g["heatmap"]).max()


def test_flip_and_candidates_fixtures(golden_dir):
    """oracle random_flip / candidate_bbox against the reference-generated vectors (make_golden_r2.py)."""
    g = np.load(os.path.join(golden_dir, "random_flip.npz"))
    pairs = g["pairs"].tolist()
    for i in range(g["images"].shape[0]):
        if g["flipped"][i]:
            im, j, v, c = onp.random_flip(g["images"][i], g["joints"][i], g["visible"][i], g["center"][i], pairs)
            assert np.array_equal(im, g["out_images"][i]) and np.array_equal(j, g["out_joints"][i])
            assert np.array_equal(v, g["out_visible"][i]) and np.array_equal(c, g["out_center"][i])
        else:
            assert np.array_equal(g["images"][i], g["out_images"][i]) and np.array_equal(g["joints"][i], g["out_joints"][i])
    c = np.load(os.path.join(golden_dir, "candidates.npz"))
    assert np.array_equal(onp.candidate_bbox(c["centre"], c["sizes"], int(c["k"]), float(c["image_size"])), c["candidates"])


def test_litehrnet_fixture(golden_dir):
    """oracle.torch_ref.LiteHRNet against the REAL reference's vectors (make_golden_r2_models.py): forward bit-for-bit,
    known parameter count 1,483,873 (test_models_performance.ipynb:276-279)."""
    g = np.load(os.path.join(golden_dir, "model_L18_128.npz"))
    m = torch_ref.get_model(litehandnet_cfg("L", depth=18))
    assert sum(p.numel() for p in m.parameters()) == 1483873
    m.load_state_dict(synth.synth_state_dict(m, int(g["seed"])))
    m.train()
    y = m(synth.synth_images(int(g["n"]), int(g["size"]), int(g["seed"])))
    assert np.abs(y.detach().numpy() - g["heatmap"]).max() <= 1e-5 * np.abs(g["heatmap"]).max()


def test_hsv_oracle_properties():
    """oracle.heatmap_np HSV pieces (random_hsv.py:20-34; colour conversion = OpenCV's 8-bit algorithm restated, parity with
    cv2 unpinned): primaries land on OpenCV's documented hues (red 0, green 60, blue 120 at hue range 180), grey has s = 0, the
    BGR -> HSV -> BGR round trip (hue has 180 levels) moves no channel by more than 6 levels, 0.6 on average, and the integer jitter wraps hue mod 180 / clips."""
    from oracle import heatmap_np as onp
    prim = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [128, 128, 128], [0, 255, 255]]], np.uint8)
    hsv = onp.bgr2hsv_u8(prim)[0]
    assert hsv[:, 0].tolist() == [0, 60, 120, 0, 30] and hsv[:, 1].tolist() == [255, 255, 255, 0, 255] and hsv[:, 2].tolist() == [255, 255, 255, 128, 255]
    r = np.random.Generator(np.random.PCG64(3))
    img = r.integers(0, 256, (48, 48, 3), dtype=np.uint8)
    back = onp.hsv2bgr_u8(onp.bgr2hsv_u8(img))
    d = np.abs(back.astype(int) - img.astype(int))
    assert d.max() <= 6 and d.mean() < 0.6
    assert np.array_equal(onp.hsv_jitter(img, [0, 0, 0]), back)
    a, b = onp.hsv_jitter(img, [5, 0, 0]), onp.hsv_jitter(img, [5 - 180, 0, 0])
    assert np.array_equal(a, b)                                       # hue wraps mod 180
    assert onp.hsv_jitter(prim, [0, 0, 30])[0, 3].tolist() == [158, 158, 158] and onp.hsv_jitter(prim, [0, 0, -300])[0].max() == 0
    np.random.seed(5)
    g = onp.hsv_gains(200)
    assert g.dtype == np.int16 and np.abs(g[:, 0]).max() <= 5 and np.abs(g[:, 1:]).max() <= 30
    assert 0.4 < (g[:, 1] == 0).mean() < 0.65                         # randint(0, 2) switches each channel off half of the time

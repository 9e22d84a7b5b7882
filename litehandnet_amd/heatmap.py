"""HIP-backed Gaussian heatmap encode / decode / metric -- mirrors of
datasets/data_pipeline/generateTarget.py (TopDownGenerateTarget, MSRA branches),
utils/post_processing/evaluation/top_down_eval.py (_get_max_preds, keypoints_from_heatmaps 'default',
keypoint_pck_accuracy), datasets/data_pipeline/post_transforms.py (transform_preds),
utils/post_processing/decoder.py (TopDownDecoder) and utils/HeatmapParser.py:41-50 (nms)."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _dev(x, device=None):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not x.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.LhnError("no GPU visible: litehandnet_amd has no CPU fallback")
        x = x.to(device or "cuda", non_blocking=True)
    return x


# ------------------------------------------------------------------ encode
def generate_target_batch(joints_3d, joints_3d_visible, image_size, heatmap_size, sigma=2, unbiased_encoding=True,
                          encoding="MSRA"):
    """[N,K,3] joints (image px) + visibility -> target [N,K,H,W], weight [N,K,1] on the GPU.
    encoding 'MSRA' (generateTarget.py:74-159, biased patch / DARK full map) or 'UDP' (:160-236, GaussianHeatmap)."""
    j = _lib.f32c(_dev(joints_3d))
    v = _lib.f32c(_dev(joints_3d_visible, j.device))
    N, K, _ = j.shape
    W, H = int(heatmap_size[0]), int(heatmap_size[1])
    target = torch.empty((N, K, H, W), dtype=torch.float32, device=j.device)
    weight = torch.empty((N, K, 1), dtype=torch.float32, device=j.device)
    L = _lib.lib()
    _lib.check(L.lhn_heatmap_encode(_lib.ptr(j), _lib.ptr(v), _lib.ptr(target), _lib.ptr(weight), N, K, H, W,
                                    C.c_float(image_size[0]), C.c_float(image_size[1]), C.c_float(sigma),
                                    2 if encoding == "UDP" else (1 if unbiased_encoding else 0), _lib.stream()),
               "lhn_heatmap_encode")
    return target, weight


class TopDownGenerateTarget:
    """generateTarget.py:34-300, MSRA encoding.  `__call__(results)` keeps the per-sample dict contract
    (numpy in, numpy out); `batch()` is the device-resident path (joints in, targets stay in HBM)."""

    def __init__(self, sigma=2, kernel=(11, 11), target_type="GaussianHeatmap", encoding="MSRA",
                 unbiased_encoding=False):
        if encoding not in ("MSRA", "UDP") or isinstance(sigma, (list, tuple)):
            raise _lib.LhnError("TopDownGenerateTarget: single-sigma MSRA / UDP encodings are built")
        if encoding == "UDP" and target_type.lower() != "gaussianheatmap":
            raise _lib.LhnError("TopDownGenerateTarget: UDP is built for target_type 'GaussianHeatmap' only")
        self.sigma, self.kernel, self.unbiased_encoding = sigma, kernel, unbiased_encoding
        self.target_type, self.encoding = target_type, encoding

    def batch(self, joints_3d, joints_3d_visible, image_size, heatmap_size):
        return generate_target_batch(joints_3d, joints_3d_visible, image_size, heatmap_size, self.sigma,
                                     self.unbiased_encoding, self.encoding)

    def __call__(self, results):
        cfg = results["ann_info"]
        if cfg.get("use_different_joint_weights", False):
            raise _lib.LhnError("use_different_joint_weights is not built")
        t, w = self.batch(np.asarray(results["joints_3d"], np.float32)[None],
                          np.asarray(results["joints_3d_visible"], np.float32)[None], cfg["image_size"],
                          cfg["heatmap_size"])
        results["target"] = t[0].cpu().numpy()
        results["target_weight"] = w[0].cpu().numpy()
        return results


# ------------------------------------------------------------------ decode
def _get_max_preds(heatmaps):
    """top_down_eval.py:199-231 on the GPU: preds [N,K,2] (x, y; -1 where max <= 0), maxvals [N,K,1]."""
    h = _lib.f32c(_dev(heatmaps))
    assert h.dim() == 4, "batch_images should be 4-ndim"
    N, K, H, W = h.shape
    preds = torch.empty((N, K, 2), dtype=torch.float32, device=h.device)
    maxvals = torch.empty((N, K, 1), dtype=torch.float32, device=h.device)
    _lib.check(_lib.lib().lhn_heatmap_argmax(_lib.ptr(h), _lib.ptr(preds), _lib.ptr(maxvals), C.c_void_p(0), N, K, H, W,
                                             _lib.stream()), "lhn_heatmap_argmax")
    return preds, maxvals


def refine_preds(heatmaps, preds, mode="default"):
    """+-0.25 shift: 'default' = top_down_eval.py:440-452; 'offset' = heatmap_post_processing.py:6-33."""
    h = _lib.f32c(_dev(heatmaps))
    p = _lib.f32c(_dev(preds, h.device)).clone()
    N, K, H, W = h.shape
    _lib.check(_lib.lib().lhn_heatmap_refine(_lib.ptr(h), _lib.ptr(p), N, K, H, W, 0 if mode == "default" else 1,
                                             _lib.stream()), "lhn_heatmap_refine")
    return p


def transform_preds(coords, center, scale, output_size, use_udp=False):
    """post_transforms.py:6-48, batched: coords [N,K,2], center/scale [N,2]."""
    c = _lib.f32c(_dev(coords))
    ce, sc = _lib.f32c(_dev(center, c.device)), _lib.f32c(_dev(scale, c.device))
    N, K, _ = c.shape
    out = torch.empty_like(c)
    _lib.check(_lib.lib().lhn_transform_preds(_lib.ptr(c), _lib.ptr(ce), _lib.ptr(sc), _lib.ptr(out), N, K,
                                              int(output_size[0]), int(output_size[1]), 1 if use_udp else 0,
                                              _lib.stream()), "lhn_transform_preds")
    return out


def keypoints_from_heatmaps(heatmaps, center, scale, post_process="default", kernel=11, use_udp=False,
                            target_type="GaussianHeatmap", only_original_preds=False):
    """top_down_eval.py:375-463 fused in one kernel (argmax -> shift -> back-transform); device tensors out."""
    if post_process not in (None, "default", "unbiased"):
        raise _lib.LhnError("keypoints_from_heatmaps: post_process None|'default'|'unbiased' are built")
    if use_udp and target_type.lower() != "gaussianheatmap":
        raise _lib.LhnError("keypoints_from_heatmaps: UDP is built for target_type 'GaussianHeatmap' only")
    h = _lib.f32c(_dev(heatmaps))
    ce, sc = _lib.f32c(_dev(center, h.device)), _lib.f32c(_dev(scale, h.device))
    N, K, H, W = h.shape
    hm_preds = torch.empty((N, K, 2), dtype=torch.float32, device=h.device)
    preds = torch.empty_like(hm_preds)
    maxvals = torch.empty((N, K, 1), dtype=torch.float32, device=h.device)
    if use_udp:      # top_down_eval.py:404-411: _get_max_preds + post_dark_udp, then the UDP back-transform
        assert kernel > 0
        _lib.check(_lib.lib().lhn_heatmap_decode_dark_udp(_lib.ptr(h), _lib.ptr(ce), _lib.ptr(sc), _lib.ptr(hm_preds),
                                                          _lib.ptr(preds), _lib.ptr(maxvals), N, K, H, W, int(kernel),
                                                          _lib.stream()), "lhn_heatmap_decode_dark_udp")
        return (preds, maxvals) if only_original_preds else (hm_preds, preds, maxvals)
    if post_process == "unbiased":
        assert kernel > 0
        _lib.check(_lib.lib().lhn_heatmap_decode_dark(_lib.ptr(h), _lib.ptr(ce), _lib.ptr(sc), _lib.ptr(hm_preds),
                                                      _lib.ptr(preds), _lib.ptr(maxvals), N, K, H, W, int(kernel),
                                                      _lib.stream()), "lhn_heatmap_decode_dark")
        return (preds, maxvals) if only_original_preds else (hm_preds, preds, maxvals)
    _lib.check(_lib.lib().lhn_heatmap_decode(_lib.ptr(h), _lib.ptr(ce), _lib.ptr(sc), _lib.ptr(hm_preds), _lib.ptr(preds),
                                             _lib.ptr(maxvals), N, K, H, W, 0 if post_process is None else 1,
                                             _lib.stream()), "lhn_heatmap_decode")
    if only_original_preds:
        return preds, maxvals
    return hm_preds, preds, maxvals


def heatmap_nms(heatmaps, kernel=11):
    """HeatmapParser.py:41-50: h * (maxpool_kxk(h) == h).  Returns a new tensor."""
    h = _lib.f32c(_dev(heatmaps)).clone()
    N, K, H, W = h.shape
    scratch = torch.empty_like(h)
    _lib.check(_lib.lib().lhn_heatmap_nms(_lib.ptr(h), _lib.ptr(scratch), N, K, H, W, int(kernel), _lib.stream()),
               "lhn_heatmap_nms")
    return h


def keypoint_pck_accuracy(pred, gt, mask, thr, normalize):
    """top_down_eval.py:129-165 -> (acc [K] tensor, avg_acc float, cnt int)."""
    p = _lib.f32c(_dev(pred))
    g = _lib.f32c(_dev(gt, p.device))
    m = _dev(mask, p.device).to(torch.uint8).contiguous()
    nz = _lib.f32c(_dev(normalize, p.device))
    N, K, _ = p.shape
    acc = torch.empty(K, dtype=torch.float32, device=p.device)
    ac = torch.empty(2, dtype=torch.float32, device=p.device)
    _lib.check(_lib.lib().lhn_pck_accuracy(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(nz), C.c_float(thr),
                                           _lib.ptr(acc), _lib.ptr(ac), N, K, _lib.stream()), "lhn_pck_accuracy")
    a = ac.cpu()
    return acc, float(a[0]), int(a[1])


def generate_simdr_batch(joints, visible, image_size, k=2, sigma=2):
    """GenerateSimDR._generate_sa_simdr (generate_simder.py:9-31) for a whole batch on the device:
    joints [N,K,3], visible [N,K,1|3] -> (simdr_x [N,K,W*k], simdr_y [N,K,H*k])."""
    j = _lib.f32c(_dev(joints))
    v = _lib.f32c(_dev(visible, j.device))
    N, K = j.shape[0], j.shape[1]
    Wd, Hd = int(image_size[0] * int(k)), int(image_size[1] * int(k))
    tx = torch.empty((N, K, Wd), dtype=torch.float32, device=j.device)
    ty = torch.empty((N, K, Hd), dtype=torch.float32, device=j.device)
    _lib.check(_lib.lib().lhn_simdr_encode(_lib.ptr(j), _lib.ptr(v), v.shape[-1], _lib.ptr(tx), _lib.ptr(ty), N, K, Wd, Hd,
                                           _lib.C.c_float(float(int(k))), _lib.C.c_float(float(sigma)), _lib.stream()),
               "lhn_simdr_encode")
    return tx, ty


class GenerateSimDR:
    """generate_simder.py:3-42 -- per-sample `results` dict contract kept; `.batch()` is the device-resident form."""

    def __init__(self, sigma=2, k=2):
        self.sigma = sigma
        self.k = int(k)
        self.with_simdr = k > 0 and not isinstance(sigma, (list, tuple))

    def batch(self, joints, visible, image_size):
        return generate_simdr_batch(joints, visible, image_size, self.k, self.sigma)

    def __call__(self, results):
        if self.with_simdr:
            tx, ty = generate_simdr_batch(torch.as_tensor(results["joints_3d"])[None], torch.as_tensor(results["joints_3d_visible"])[None],
                                          results["ann_info"]["image_size"], self.k, self.sigma)
            results["simdr_x"], results["simdr_y"] = tx[0], ty[0]
        return results


def keypoints_from_simdr(x_vectors, y_vectors, center, scale, k=2):
    """top_down_eval.py:466-500: argmax of each 1-D vector / k, score = mean of the two maxima, back-transform with the
    un-scaled map size.  Returns [N,K,3] on the device."""
    assert k > 0
    xv = _lib.f32c(_dev(x_vectors))
    yv = _lib.f32c(_dev(y_vectors, xv.device))
    N, K, Wd = xv.shape
    Hd = yv.shape[2]
    def argmax1d(v, L):
        idx = torch.empty((N, K), dtype=torch.int32, device=v.device)
        mv = torch.empty((N, K, 1), dtype=torch.float32, device=v.device)
        scratch = torch.empty((N, K, 2), dtype=torch.float32, device=v.device)
        _lib.check(_lib.lib().lhn_heatmap_argmax(_lib.ptr(v), _lib.ptr(scratch), _lib.ptr(mv), _lib.ptr(idx), N, K, 1, L,
                                                 _lib.stream()), "lhn_heatmap_argmax")
        return idx.to(torch.float32).unsqueeze(2), mv      # plain first-maximum index (no "-1 where max <= 0" masking)
    ix, mx = argmax1d(xv, Wd)
    iy, my = argmax1d(yv, Hd)
    preds = torch.cat([ix, iy], dim=2) / float(k)
    out = transform_preds(preds, center, scale, [Wd // k, Hd // k])
    return torch.cat([out, (mx + my) / 2], dim=2)


def candidate_bbox(center_maps, size_maps, num_candidates, image_size):
    """HeatmapParser.candidate_bbox (utils/HeatmapParser.py:52-85): center_maps [N,H,W] (peak-suppressed, see heatmap_nms),
    size_maps [N,2,H,W] already region-averaged (or None) -> candidates [N,k,5] = (x, y, w, h, confidence), descending."""
    cm = _lib.f32c(_dev(center_maps))
    N, H, W = cm.shape
    sm = None if size_maps is None else _lib.f32c(_dev(size_maps, cm.device))
    out = torch.empty((N, int(num_candidates), 5), dtype=torch.float32, device=cm.device)
    _lib.check(_lib.lib().lhn_heatmap_topk(_lib.ptr(cm), _lib.ptr(sm), _lib.ptr(out), N, H, W, int(num_candidates),
                                           C.c_float(float(image_size)), _lib.stream()), "lhn_heatmap_topk")
    return out


class TopDownDecoder:
    """utils/post_processing/decoder.py:9-107.  Same attributes (`k` is read by test.py:125) and the same result dicts: host
    numpy arrays for preds / hm_preds / boxes / output_heatmap, a python list for bbox_ids.  The decode itself runs on the
    device; everything small comes back in ONE device-to-host copy.  `as_numpy=False` keeps the results as device tensors
    (the heat maps then never leave HBM -- the reference's evaluate() only reads preds / boxes / image_paths / bbox_ids)."""

    def __init__(self, cfg, as_numpy=True):
        self.image_size = np.array(cfg.DATASET.image_size)
        self.heatmap_size = np.array(cfg.DATASET.heatmap_size)
        self.num_joints = cfg.DATASET.num_joints
        self.post_process = "unbiased" if cfg.PIPELINE.unbiased_encoding else "default"
        self.kernel = cfg.PIPELINE.kernel[0]
        self.use_udp = cfg.PIPELINE.use_udp
        self.k = cfg.PIPELINE.get("simdr_split_ratio", 0)
        self.as_numpy = as_numpy

    def _boxes(self, meta, center, scale, dev):
        n = center.shape[0]
        b = torch.zeros((n, 6), dtype=torch.float32, device=dev)
        b[:, 0:2] = center[:, 0:2]
        b[:, 2:4] = scale[:, 0:2]
        b[:, 4] = torch.prod(scale * 200.0, dim=1)
        b[:, 5] = _dev(meta["bbox_score"], dev).float().reshape(-1)
        return b

    @staticmethod
    def _ids(meta):
        ids = meta["bbox_id"]
        return ids.tolist() if hasattr(ids, "tolist") else list(ids)

    def _finish(self, res, small, out):
        """small: list of (key, [N, ..] device tensor) packed into one host copy; out: the heat maps."""
        if not self.as_numpy:
            res.update(dict(small), output_heatmap=out)
            return res
        n = small[0][1].shape[0]
        flat = torch.cat([t.reshape(n, -1) for _, t in small], dim=1).cpu().numpy()
        off = 0
        for k, t in small:
            w = t[0].numel()
            res[k] = np.ascontiguousarray(flat[:, off:off + w]).reshape(tuple(t.shape))
            off += w
        res["output_heatmap"] = out.detach().cpu().numpy()
        return res

    def decode(self, meta, model_output, post_process=None):
        pp = post_process or self.post_process
        out = model_output[:, :self.num_joints]
        center, scale = _dev(meta["center"], out.device).float(), _dev(meta["scale"], out.device).float()
        hm_preds, preds, maxvals = keypoints_from_heatmaps(out, center, scale, post_process=pp, kernel=self.kernel,
                                                           use_udp=self.use_udp)
        res = dict(image_paths=meta.get("image_file"), bbox_ids=self._ids(meta))
        small = [("preds", torch.cat([preds, maxvals], dim=2)), ("hm_preds", torch.cat([hm_preds * 4, maxvals], dim=2)),
                 ("boxes", self._boxes(meta, center, scale, out.device))]
        res = self._finish(res, small, out)
        return {k: res[k] for k in ("preds", "hm_preds", "boxes", "image_paths", "bbox_ids", "output_heatmap")}

    def decode_simdr(self, meta, model_output):
        """decoder.py:73-107: keypoints from the SimDR vectors in `meta`, boxes as in decode()."""
        out = model_output[:, :self.num_joints]
        center, scale = _dev(meta["center"], out.device).float(), _dev(meta["scale"], out.device).float()
        preds = keypoints_from_simdr(_dev(meta["simdr_x"], out.device), _dev(meta["simdr_y"], out.device), center, scale, self.k)
        res = dict(image_paths=meta.get("image_file"), bbox_ids=self._ids(meta))
        res = self._finish(res, [("preds", preds), ("boxes", self._boxes(meta, center, scale, out.device))], out)
        return {k: res[k] for k in ("preds", "boxes", "image_paths", "bbox_ids", "output_heatmap")}

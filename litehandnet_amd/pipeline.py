"""GPU input path: the per-sample CPU transforms of datasets/build_dataset.py:111-132 that feed
the network -- TopDownAffine (topdown_affine.py:47-114, non-UDP) + ToTensor + NormalizeTensor (shared_transform.py:3-44)
-- as ONE launch over a batch, followed by the device-resident target encoders in `heatmap` (TopDownGenerateTarget,
GenerateSimDR).  Source images of one batch share their size (uint8, HWC)."""
import ctypes as C

import numpy as np
import torch

from . import _lib, heatmap

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def hsv_gains(n, hue_delta=5, saturation_delta=30, value_delta=30):
    """The random draws of HSVRandomAug.__call__ (random_hsv.py:22-29) for n samples: numpy's global generator, the reference's
    calls in the reference's order (so `np.random.seed` reproduces the reference's stream) -> int16 [n, 3]."""
    out = np.zeros((n, 3), np.int16)
    for i in range(n):
        g = np.random.uniform(-1, 1, 3) * [hue_delta, saturation_delta, value_delta]
        g *= np.random.randint(0, 2, 3)
        out[i] = g.astype(np.int16)
    return out


def hsv_jitter(images_u8, gains):
    """HSVRandomAug on a device batch of uint8 BGR images [N,H,W,3] (a copy is jittered and returned)."""
    _lib.require_device()
    dev = torch.device("cuda", torch.cuda.current_device())
    img = torch.as_tensor(images_u8).to(dev)
    if img.dtype != torch.uint8 or img.dim() != 4 or img.shape[3] != 3:
        raise _lib.LhnError("hsv_jitter: images must be uint8 [N,H,W,3]")
    img = img.contiguous().clone()
    g = torch.as_tensor(np.asarray(gains, np.int16)).to(dev).contiguous()
    if tuple(g.shape) != (img.shape[0], 3):
        raise _lib.LhnError("hsv_jitter: gains must be int16 [N,3]")
    _lib.check(_lib.lib().lhn_hsv_jitter(_lib.ptr(img), _lib.ptr(g), int(img.shape[0]), int(img.shape[1]), int(img.shape[2]),
                                         _lib.stream()), "lhn_hsv_jitter")
    return img


def random_scale_rotation(scale, rot_factor=40, scale_factor=0.5, rot_prob=0.6):
    """TopDownGetRandomScaleRotation.__call__ (topdown_affine.py:28-45) for a batch: three draws per sample from numpy's global
    generator in the reference's order (randn, randn, rand) -> (scale * s_factor [n,2], rotation [n] degrees).  Host side: three
    numbers per sample; the arithmetic on the image is lhn_affine_warp_normalize."""
    scale = np.asarray(scale)
    out_s, out_r = np.empty(scale.shape, np.float64), np.empty(scale.shape[0], np.float64)
    for i in range(scale.shape[0]):
        s_factor = np.clip(np.random.randn() * scale_factor + 1, 1 - scale_factor, 1 + scale_factor)
        r_factor = np.clip(np.random.randn() * rot_factor, -rot_factor * 2, rot_factor * 2)
        out_r[i] = r_factor if np.random.rand() <= rot_prob else 0
        out_s[i] = scale[i] * s_factor
    return out_s, out_r


def random_flip(joints, visible, center, flipped, flip_pairs, img_width):
    """TopDownRandomFlip / fliplr_joints (RandomFlip.py:28-100) for a device-resident batch: returns NEW (joints, visible,
    center) with the flagged samples flipped.  The image itself is not touched: pass the same `flipped` flags to
    affine_warp_normalize, which reads those source images mirrored."""
    _lib.require_device()
    dev = torch.device("cuda", torch.cuda.current_device())
    j = _lib.f32c(torch.as_tensor(joints, dtype=torch.float32).to(dev)).clone()
    v = _lib.f32c(torch.as_tensor(visible, dtype=torch.float32).to(dev)).clone()
    c = _lib.f32c(torch.as_tensor(center, dtype=torch.float32).to(dev)).clone().reshape(-1, 2)
    f = torch.as_tensor(flipped).to(dev).to(torch.uint8).contiguous()
    N, K = j.shape[0], j.shape[1]
    pr = torch.as_tensor(list(flip_pairs), dtype=torch.int32).reshape(-1, 2).to(dev).contiguous()
    _lib.check(_lib.lib().lhn_random_flip(_lib.ptr(j), _lib.ptr(v), int(v.shape[-1]), _lib.ptr(c), _lib.ptr(f),
                                          _lib.ptr(pr) if pr.numel() else None, int(pr.shape[0]), N, K, int(img_width),
                                          _lib.stream()), "lhn_random_flip")
    return j, v, c


def affine_warp_normalize(images_u8, center, scale, rotation, image_size, joints=None, visible=None, mean=IMAGENET_MEAN,
                          std=IMAGENET_STD, use_udp=False, flipped=None):
    """images_u8 [N,Hs,Ws,3] uint8, center/scale [N,2], rotation [N] degrees -> float32 [N,3,H,W] normalised crops
    (+ joints [N,K,3] mapped into the crop, visible ones only, when given).  flipped [N]: those source images are read
    mirrored (the image half of TopDownRandomFlip; joints / center come from random_flip)."""
    _lib.require_device()
    dev = torch.device("cuda", torch.cuda.current_device())
    img = torch.as_tensor(images_u8).to(dev)
    if img.dtype != torch.uint8 or img.dim() != 4 or img.shape[3] != 3:
        raise _lib.LhnError("affine_warp_normalize: images must be uint8 [N,Hs,Ws,3]")
    img = img.contiguous()
    N, Hs, Ws, _ = img.shape
    ce = _lib.f32c(torch.as_tensor(center, dtype=torch.float32).to(dev)).reshape(N, 2)
    sc = _lib.f32c(torch.as_tensor(scale, dtype=torch.float32).to(dev)).reshape(N, 2)
    ro = _lib.f32c(torch.as_tensor(rotation, dtype=torch.float32).to(dev)).reshape(N)
    Wo, Ho = int(image_size[0]), int(image_size[1])
    out = torch.empty((N, 3, Ho, Wo), dtype=torch.float32, device=dev)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    j = v = None
    K = 0
    if joints is not None:
        j = _lib.f32c(torch.as_tensor(joints, dtype=torch.float32).to(dev)).clone()
        v = _lib.f32c(torch.as_tensor(visible, dtype=torch.float32).to(dev))
        K = j.shape[1]
    fl = None if flipped is None else torch.as_tensor(flipped).to(dev).to(torch.uint8).contiguous()
    _lib.check(_lib.lib().lhn_affine_warp_normalize2(_lib.ptr(img), N, Hs, Ws, _lib.ptr(ce), _lib.ptr(sc), _lib.ptr(ro), m3, s3,
                                                     _lib.ptr(out), Ho, Wo, _lib.ptr(j), _lib.ptr(v),
                                                     0 if v is None else v.shape[-1], K, 1 if use_udp else 0, _lib.ptr(fl),
                                                     _lib.stream()),
               "lhn_affine_warp_normalize")
    return (out, j) if joints is not None else out


class TopDownBatchPipeline:
    """The evaluation pipeline of build_dataset.py:124-131 for a batch already resident on the GPU: affine crop +
    normalise, then targets (and SimDR vectors when simdr_split_ratio > 0) from the transformed joints."""

    def __init__(self, cfg):
        P = cfg.PIPELINE
        self.use_udp, self.encoding = bool(P.use_udp), P.get("encoding", "MSRA")
        self.image_size = list(cfg.DATASET.image_size)
        self.heatmap_size = list(cfg.DATASET.heatmap_size)
        self.sigma, self.unbiased, self.k = P.sigma, bool(P.unbiased_encoding), int(P.get("simdr_split_ratio", 0))

    def __call__(self, images_u8, center, scale, rotation, joints, visible):
        img, j = affine_warp_normalize(images_u8, center, scale, rotation, self.image_size, joints, visible, use_udp=self.use_udp)
        vis = torch.as_tensor(visible, dtype=torch.float32).to(img.device)
        target, weight = heatmap.generate_target_batch(j, vis, self.image_size, self.heatmap_size, self.sigma, self.unbiased,
                                                       self.encoding)
        meta = {"target": target, "target_weight": weight, "joints_3d": j}
        if self.k > 0:
            meta["simdr_x"], meta["simdr_y"] = heatmap.generate_simdr_batch(j, vis, self.image_size, self.k, self.sigma)
        return img, meta


class TopDownTrainPipeline(TopDownBatchPipeline):
    """The TRAINING pipeline of build_dataset.py:111-122 for a batch resident on the GPU, in the reference's order: HSVRandomAug
    -> TopDownRandomFlip -> TopDownGetRandomScaleRotation -> TopDownAffine -> ToTensor / NormalizeTensor -> targets.  The random
    numbers come from numpy's global generator on the host, drawn per transform for the whole batch (the reference draws them
    per sample inside DataLoader workers, so the streams differ by construction; each transform's draws are the reference's)."""

    def __init__(self, cfg, flip_pairs=()):
        super().__init__(cfg)
        P = cfg.PIPELINE
        self.flip_prob = float(P.get("flip_prob", 0.5))
        self.rot_factor, self.scale_factor, self.rot_prob = P.get("rot_factor", 40), P.get("scale_factor", 0.5), P.get("rot_prob", 0.6)
        self.flip_pairs = list(flip_pairs)

    def __call__(self, images_u8, center, scale, joints, visible):
        n = int(torch.as_tensor(images_u8).shape[0])
        img = hsv_jitter(images_u8, hsv_gains(n))
        flipped = np.random.rand(n) <= self.flip_prob                       # RandomFlip.py:45
        j, v, c = random_flip(joints, visible, center, flipped, self.flip_pairs, int(img.shape[2]))
        s, r = random_scale_rotation(np.asarray(torch.as_tensor(scale).cpu()), self.rot_factor, self.scale_factor, self.rot_prob)
        out, jj = affine_warp_normalize(img, c, s.astype(np.float32), r.astype(np.float32), self.image_size, j, v, use_udp=self.use_udp,
                                        flipped=flipped)
        target, weight = heatmap.generate_target_batch(jj, v, self.image_size, self.heatmap_size, self.sigma, self.unbiased, self.encoding)
        meta = {"target": target, "target_weight": weight, "joints_3d": jj, "rotation": r, "scale": s, "flipped": flipped}
        if self.k > 0:
            meta["simdr_x"], meta["simdr_y"] = heatmap.generate_simdr_batch(jj, v, self.image_size, self.k, self.sigma)
        return out, meta

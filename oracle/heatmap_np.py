"""ORACLE (test infrastructure only -- never imported by the product path).

numpy restatement of the reference's Gaussian heatmap encode / decode / metric
arithmetic.  Pinned against the real reference functions (loaded by file path in
the build container) by `tests/golden/make_golden.py`; fixtures in tests/golden/.

Reference behaviour followed (paths relative to the reference tree):
  * encode   datasets/data_pipeline/generateTarget.py:74-159  (_msra_generate_target)
  * argmax   utils/post_processing/evaluation/top_down_eval.py:199-231 (_get_max_preds)
  * shift    utils/post_processing/evaluation/top_down_eval.py:440-452 ('default' branch)
  * shift'   utils/heatmap_post_processing.py:6-33 (legacy twin: clamped neighbours, +0.5)
  * DARK     utils/post_processing/evaluation/top_down_eval.py:233-272, 338-372, 433-439
  * back-map datasets/data_pipeline/post_transforms.py:6-48 (transform_preds)
  * NMS      utils/HeatmapParser.py:41-50 (11x11 max-pool peak keep)
  * PCK/AUC/EPE  utils/post_processing/evaluation/top_down_eval.py:12-62,104-196
DARK's blur uses cv2.GaussianBlur in the reference; cv2 is absent, so the blur
here follows OpenCV's documented kernel rule and is "parity unpinned" at bit level
(see DESIGN.md).
"""
import numpy as np


# ----------------------------------------------------------------- encode
def msra_generate_target(joints_3d, joints_3d_visible, image_size, heatmap_size, sigma=2,
                         unbiased=True):
    """joints_3d [K,3] image coords, visible [K,3] -> target [K,H,W] f32, weight [K,1] f32."""
    K = joints_3d.shape[0]
    W, H = heatmap_size
    image_size = np.asarray(image_size)
    weight = np.zeros((K, 1), np.float32)
    target = np.zeros((K, H, W), np.float32)
    r = sigma * 3
    stride = image_size / [W, H]                       # float64, as in the reference
    xs = np.arange(0, W, 1, np.float32)
    ys = np.arange(0, H, 1, np.float32)[:, None]
    for k in range(K):
        weight[k] = joints_3d_visible[k, 0]
        if unbiased:
            mx = joints_3d[k][0] / stride[0]
            my = joints_3d[k][1] / stride[1]
            if mx - r >= W or my - r >= H or mx + r + 1 < 0 or my + r + 1 < 0:
                weight[k] = 0
            if weight[k] > 0.5:
                target[k] = np.exp(-((xs - mx) ** 2 + (ys - my) ** 2) / (2 * sigma ** 2))
        else:
            mx = int(joints_3d[k][0] / stride[0] + 0.5)
            my = int(joints_3d[k][1] / stride[1] + 0.5)
            ul = [int(mx - r), int(my - r)]
            br = [int(mx + r + 1), int(my + r + 1)]
            if ul[0] >= W or ul[1] >= H or br[0] < 0 or br[1] < 0:
                weight[k] = 0
            if weight[k] > 0.5:
                size = 2 * r + 1
                g1 = np.arange(0, size, 1, np.float32)
                g = np.exp(-((g1 - size // 2) ** 2 + (g1[:, None] - size // 2) ** 2) / (2 * sigma ** 2))
                gx = max(0, -ul[0]), min(br[0], W) - ul[0]
                gy = max(0, -ul[1]), min(br[1], H) - ul[1]
                ix = max(0, ul[0]), min(br[0], W)
                iy = max(0, ul[1]), min(br[1], H)
                target[k][iy[0]:iy[1], ix[0]:ix[1]] = g[gy[0]:gy[1], gx[0]:gx[1]]
    return target, weight


# ----------------------------------------------------------------- decode
def get_max_preds(heatmaps):
    """[N,K,H,W] -> preds [N,K,2] f32 (x = idx % W, y = idx // W; -1 where max <= 0), maxvals [N,K,1]."""
    N, K, _, W = heatmaps.shape
    flat = heatmaps.reshape(N, K, -1)
    idx = flat.argmax(2)
    maxvals = flat.max(2)[..., None]
    preds = np.stack([idx % W, idx // W], -1).astype(np.float32)
    preds = np.where(maxvals > 0.0, preds, np.float32(-1))
    return preds.astype(np.float32), maxvals


def refine_default(heatmaps, preds):
    """+-0.25 sign shift, only strictly inside the border (1 < p < size-1)."""
    N, K, H, W = heatmaps.shape
    out = preds.copy()
    for n in range(N):
        for k in range(K):
            px, py = int(out[n, k, 0]), int(out[n, k, 1])
            if 1 < px < W - 1 and 1 < py < H - 1:
                h = heatmaps[n, k]
                d = np.array([h[py, px + 1] - h[py, px - 1], h[py + 1, px] - h[py - 1, px]])
                out[n, k] += np.sign(d) * 0.25
    return out


def refine_offset_legacy(heatmaps, kpts_xy):
    """utils/heatmap_post_processing.py:6-33: clamped neighbours, strict '>' else minus, then +0.5."""
    N, K, H, W = heatmaps.shape
    out = kpts_xy.astype(np.float32).copy()
    for n in range(N):
        for k in range(K):
            x, y = out[n, k]
            xx, yy = int(x), int(y)
            h = heatmaps[n, k]
            x += 0.25 if h[yy, min(xx + 1, W - 1)] > h[yy, max(xx - 1, 0)] else -0.25
            y += 0.25 if h[min(yy + 1, H - 1), xx] > h[max(yy - 1, 0), xx] else -0.25
            out[n, k] = (x + 0.5, y + 0.5)
    return out


def transform_preds(coords, center, scale, output_size, use_udp=False):
    scale = scale * 200.0
    if use_udp:
        sx, sy = scale[0] / (output_size[0] - 1.0), scale[1] / (output_size[1] - 1.0)
    else:
        sx, sy = scale[0] / output_size[0], scale[1] / output_size[1]
    out = np.ones_like(coords)
    out[:, 0] = coords[:, 0] * sx + center[0] - scale[0] * 0.5
    out[:, 1] = coords[:, 1] * sy + center[1] - scale[1] * 0.5
    return out


def opencv_gaussian_kernel1d(ksize):
    """cv2.getGaussianKernel(ksize, sigma<=0): sigma = 0.3*((ksize-1)*0.5-1)+0.8, normalised.
    (OpenCV uses fixed tables for ksize<=7 with sigma<=0; 11 and 19 use the formula.)"""
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2 * sigma * sigma))
    return (k / k.sum())


def gaussian_blur(heatmaps, kernel=11):
    """Separable blur of the zero-padded map then max re-normalisation (top_down_eval.py:233-272)."""
    k1 = opencv_gaussian_kernel1d(kernel).astype(np.float32)
    b = (kernel - 1) // 2
    out = np.empty_like(heatmaps)
    N, K, H, W = heatmaps.shape
    for n in range(N):
        for j in range(K):
            src = heatmaps[n, j]
            omax = src.max()
            pad = np.zeros((H + 2 * b, W + 2 * b), np.float32)
            pad[b:-b, b:-b] = src
            tmp = np.zeros_like(pad)
            for t in range(kernel):           # rows (x direction) first, zero border
                sh = t - b
                lo, hi = max(0, -sh), min(pad.shape[1], pad.shape[1] - sh)
                tmp[:, lo:hi] += k1[t] * pad[:, lo + sh:hi + sh]
            res = np.zeros_like(pad)
            for t in range(kernel):
                sh = t - b
                lo, hi = max(0, -sh), min(pad.shape[0], pad.shape[0] - sh)
                res[lo:hi, :] += k1[t] * tmp[lo + sh:hi + sh, :]
            res = res[b:-b, b:-b]
            out[n, j] = res * (omax / res.max())
    return out


def taylor(hm, coord):
    H, W = hm.shape
    px, py = int(coord[0]), int(coord[1])
    if 1 < px < W - 2 and 1 < py < H - 2:
        dx = 0.5 * (hm[py][px + 1] - hm[py][px - 1])
        dy = 0.5 * (hm[py + 1][px] - hm[py - 1][px])
        dxx = 0.25 * (hm[py][px + 2] - 2 * hm[py][px] + hm[py][px - 2])
        dxy = 0.25 * (hm[py + 1][px + 1] - hm[py - 1][px + 1] - hm[py + 1][px - 1] + hm[py - 1][px - 1])
        dyy = 0.25 * (hm[py + 2][px] - 2 * hm[py][px] + hm[py - 2][px])
        det = dxx * dyy - dxy ** 2
        if det != 0:
            inv = np.linalg.inv(np.array([[dxx, dxy], [dxy, dyy]]))
            coord = coord + (-inv @ np.array([dx, dy]))
    return coord


def keypoints_from_heatmaps(heatmaps, center, scale, post_process="default", kernel=11):
    """-> (hm_preds [N,K,2], preds [N,K,2] image coords, maxvals [N,K,1]); use_udp=False path."""
    heatmaps = heatmaps.copy()
    N, K, H, W = heatmaps.shape
    hm_preds, maxvals = get_max_preds(heatmaps)
    if post_process == "unbiased":
        lg = np.log(np.maximum(gaussian_blur(heatmaps, kernel), 1e-10))
        for n in range(N):
            for k in range(K):
                hm_preds[n, k] = taylor(lg[n, k], hm_preds[n, k])
    elif post_process is not None:
        hm_preds = refine_default(heatmaps, hm_preds)
    preds = hm_preds.copy()
    for i in range(N):
        preds[i] = transform_preds(preds[i], center[i], scale[i], [W, H])
    return hm_preds, preds, maxvals


def heatmap_nms(heatmaps, kernel=11):
    """h * (maxpool_{k x k, stride 1, pad k//2}(h) == h)   (HeatmapParser.py:41-50; -inf padding)."""
    N, K, H, W = heatmaps.shape
    p = kernel // 2
    pad = np.full((N, K, H + 2 * p, W + 2 * p), -np.inf, heatmaps.dtype)
    pad[:, :, p:p + H, p:p + W] = heatmaps
    mx = np.full_like(heatmaps, -np.inf)
    for dy in range(kernel):
        for dx in range(kernel):
            mx = np.maximum(mx, pad[:, :, dy:dy + H, dx:dx + W])
    return heatmaps * (mx == heatmaps).astype(heatmaps.dtype)


# ----------------------------------------------------------------- metrics
def _calc_distances(preds, targets, mask, normalize):
    N, K, _ = preds.shape
    m = mask.copy()
    m[np.where((normalize == 0).sum(1))[0], :] = False
    d = np.full((N, K), -1, np.float32)
    normalize = normalize.copy()
    normalize[np.where(normalize <= 0)] = 1e6
    d[m] = np.linalg.norm(((preds - targets) / normalize[:, None, :])[m], axis=-1)
    return d.T


def _distance_acc(d, thr):
    v = d != -1
    n = v.sum()
    return (d[v] < thr).sum() / n if n > 0 else -1


def keypoint_pck_accuracy(pred, gt, mask, thr, normalize):
    d = _calc_distances(pred, gt, mask, normalize)
    acc = np.array([_distance_acc(x, thr) for x in d])
    valid = acc[acc >= 0]
    return acc, (valid.mean() if len(valid) else 0), len(valid)


def keypoint_auc(pred, gt, mask, normalize, num_step=20):
    nor = np.tile(np.array([[normalize, normalize]]), (pred.shape[0], 1))
    ys = [keypoint_pck_accuracy(pred, gt, mask, i / num_step, nor)[1] for i in range(num_step)]
    return float(sum(ys) / num_step)


def keypoint_epe(pred, gt, mask):
    d = _calc_distances(pred, gt, mask, np.ones((pred.shape[0], pred.shape[2]), np.float32))
    v = d[d != -1]
    return v.sum() / max(1, len(v))


def generate_sa_simdr(joints, target_weight, image_size, k=2, sigma=2):
    """datasets/data_pipeline/generate_simder.py:9-31 for one sample: joints [K,3], target_weight [K,1|3]."""
    k = int(k)
    K = joints.shape[0]
    tx = np.zeros((K, int(image_size[0] * k)), np.float32)
    ty = np.zeros((K, int(image_size[1] * k)), np.float32)
    for j in range(K):
        if target_weight[j][0] > 0:
            mu_x, mu_y = joints[j, :2] * k
            x = np.arange(0, int(image_size[0] * k), 1, np.float32)
            y = np.arange(0, int(image_size[1] * k), 1, np.float32)
            tx[j] = np.exp(-((x - mu_x) ** 2) / (2 * sigma ** 2))
            ty[j] = np.exp(-((y - mu_y) ** 2) / (2 * sigma ** 2))
    return tx, ty


def keypoints_from_simdr(x_vectors, y_vectors, center, scale, k=2):
    """utils/post_processing/evaluation/top_down_eval.py:466-500."""
    B, K, W = x_vectors.shape
    H = y_vectors.shape[2]
    preds = np.zeros((B, K, 2), np.float32)
    scores = np.zeros((B, K, 1), np.float32)
    for i in range(B):
        xi, yi = x_vectors[i].argmax(1).reshape(-1, 1), y_vectors[i].argmax(1).reshape(-1, 1)
        preds[i] = np.concatenate([xi, yi], 1) / k
        scores[i] = (x_vectors[i].max(1).reshape(-1, 1) + y_vectors[i].max(1).reshape(-1, 1)) / 2
    for i in range(B):
        preds[i] = transform_preds(preds[i], center[i], scale[i], [W // k, H // k], use_udp=False)
    return np.concatenate([preds, scores], 2)


def affine_matrix(center, scale, rot, output_size):
    """Closed form of post_transforms.get_affine_transform (:101-156; the reference solves three point pairs with
    cv2.getAffineTransform): dst = s * R(-rot) * (src - center) + output_size / 2, s = out_w / (scale_x * 200)."""
    s = output_size[0] / (scale[0] * 200.0)
    r = np.pi * rot / 180.0
    c, sn = np.cos(r), np.sin(r)
    A = s * np.array([[c, sn], [-sn, c]], np.float64)
    t = np.array([output_size[0] * 0.5, output_size[1] * 0.5]) - A @ np.asarray(center, np.float64)
    return np.concatenate([A, t[:, None]], 1)


def warp_matrix_udp(theta, size_input, size_dst, size_target):
    """post_transforms.get_warp_matrix (:49-83), float64."""
    th = np.deg2rad(theta)
    sx, sy = size_dst[0] / size_target[0], size_dst[1] / size_target[1]
    c, s = np.cos(th), np.sin(th)
    return np.array([[c * sx, -s * sx, sx * (-0.5 * size_input[0] * c + 0.5 * size_input[1] * s + 0.5 * size_target[0])],
                     [s * sy, c * sy, sy * (-0.5 * size_input[0] * s - 0.5 * size_input[1] * c + 0.5 * size_target[1])]], np.float64)


def warp_affine_normalize(img_u8, center, scale, rot, output_size, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                          use_udp=False):
    """TopDownAffine (topdown_affine.py:95-107) + ToTensor + NormalizeTensor for one HWC uint8 image, with exact bilinear
    interpolation (cv2 interpolates with 5-bit fixed-point weights -- parity unpinned, cv2 absent) and uint8 rounding."""
    M = affine_matrix(center, scale, rot, output_size)
    if use_udp:       # topdown_affine.py:76-93
        M = warp_matrix_udp(rot, np.asarray(center, np.float64) * 2.0, np.asarray(output_size, np.float64) - 1.0,
                            np.asarray(scale, np.float64) * 200.0)
    Ai = np.linalg.inv(M[:, :2])
    W, H = int(output_size[0]), int(output_size[1])
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    d = np.stack([xs - M[0, 2], ys - M[1, 2]], -1) @ Ai.T
    sx, sy = d[..., 0], d[..., 1]
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    ax, ay = sx - x0, sy - y0
    Hs, Ws = img_u8.shape[:2]
    out = np.zeros((H, W, 3), np.float64)
    for j in (0, 1):
        for i in (0, 1):
            xx, yy = x0 + i, y0 + j
            ok = (xx >= 0) & (xx < Ws) & (yy >= 0) & (yy < Hs)
            wgt = (ax if i else 1 - ax) * (ay if j else 1 - ay)
            out += (wgt * ok)[..., None] * img_u8[np.clip(yy, 0, Hs - 1), np.clip(xx, 0, Ws - 1)].astype(np.float64)
    u8 = np.clip(np.rint(out), 0, 255)
    return (((u8 / 255.0) - np.asarray(mean)) / np.asarray(std)).transpose(2, 0, 1).astype(np.float32), M


def udp_generate_target(joints_3d, joints_3d_visible, image_size, heatmap_size, sigma=2):
    """datasets/data_pipeline/generateTarget.py:160-236 (_udp_generate_target, GaussianHeatmap) for one sample."""
    image_size, heatmap_size = np.asarray(image_size), np.asarray(heatmap_size)
    K = joints_3d.shape[0]
    W, H = int(heatmap_size[0]), int(heatmap_size[1])
    weight = np.ones((K, 1), np.float32)
    weight[:, 0] = joints_3d_visible[:, 0]
    target = np.zeros((K, H, W), np.float32)
    tmp = sigma * 3
    size = 2 * tmp + 1
    x = np.arange(0, size, 1, np.float32)
    y = x[:, None]
    for j in range(K):
        fs = (image_size - 1.0) / (heatmap_size - 1.0)
        mu_x, mu_y = int(joints_3d[j][0] / fs[0] + 0.5), int(joints_3d[j][1] / fs[1] + 0.5)
        ul = [int(mu_x - tmp), int(mu_y - tmp)]
        br = [int(mu_x + tmp + 1), int(mu_y + tmp + 1)]
        if ul[0] >= W or ul[1] >= H or br[0] < 0 or br[1] < 0:
            weight[j] = 0
            continue
        x0 = y0 = size // 2
        x0 = x0 + joints_3d[j][0] / fs[0] - mu_x
        y0 = y0 + joints_3d[j][1] / fs[1] - mu_y
        g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
        gx = max(0, -ul[0]), min(br[0], W) - ul[0]
        gy = max(0, -ul[1]), min(br[1], H) - ul[1]
        ix = max(0, ul[0]), min(br[0], W)
        iy = max(0, ul[1]), min(br[1], H)
        if weight[j] > 0.5:
            target[j][iy[0]:iy[1], ix[0]:ix[1]] = g[gy[0]:gy[1], gx[0]:gx[1]]
    return target, weight


def gaussian_blur_reflect(hm, k):
    """cv2.GaussianBlur(hm, (k, k), 0) with its default BORDER_REFLECT_101, restated (cv2 absent: parity unpinned)."""
    k1 = opencv_gaussian_kernel1d(k)
    b = (k - 1) // 2
    p = np.pad(hm.astype(np.float32), b, mode="reflect")
    H, W = hm.shape
    tmp = np.zeros((H + 2 * b, W), np.float32)
    for t in range(k):
        tmp += k1[t] * p[:, t:t + W]
    out = np.zeros((H, W), np.float32)
    for t in range(k):
        out += k1[t] * tmp[t:t + H]
    return out


def post_dark_udp(coords, batch_heatmaps, kernel=3):
    """utils/post_processing/evaluation/top_down_eval.py:275-337."""
    batch_heatmaps = batch_heatmaps.copy()
    B, K, H, W = batch_heatmaps.shape
    N = coords.shape[0]
    for b in range(B):
        for k in range(K):
            batch_heatmaps[b, k] = gaussian_blur_reflect(batch_heatmaps[b, k], kernel)
    np.clip(batch_heatmaps, 0.001, 50, batch_heatmaps)
    np.log(batch_heatmaps, batch_heatmaps)
    pad = np.pad(batch_heatmaps, ((0, 0), (0, 0), (1, 1), (1, 1)), mode="edge").flatten()
    index = coords[..., 0] + 1 + (coords[..., 1] + 1) * (W + 2)
    index += (W + 2) * (H + 2) * np.arange(0, B * K).reshape(-1, K)
    index = index.astype(int).reshape(-1, 1)
    i_, ix1, iy1 = pad[index], pad[index + 1], pad[index + W + 2]
    ix1y1, ix1_y1_ = pad[index + W + 3], pad[index - W - 3]
    ix1_, iy1_ = pad[index - 1], pad[index - 2 - W]
    dx, dy = 0.5 * (ix1 - ix1_), 0.5 * (iy1 - iy1_)
    derivative = np.concatenate([dx, dy], axis=1).reshape(N, K, 2, 1)
    dxx, dyy = ix1 - 2 * i_ + ix1_, iy1 - 2 * i_ + iy1_
    dxy = 0.5 * (ix1y1 - ix1 - iy1 + i_ + i_ - ix1_ - iy1_ + ix1_y1_)
    hessian = np.concatenate([dxx, dxy, dxy, dyy], axis=1).reshape(N, K, 2, 2)
    hessian = np.linalg.inv(hessian + np.finfo(np.float32).eps * np.eye(2))
    coords = coords - np.einsum("ijmn,ijnk->ijmk", hessian, derivative).squeeze()
    return coords.astype(np.float32)


def keypoints_from_heatmaps_udp(heatmaps, center, scale, kernel=11):
    """keypoints_from_heatmaps(use_udp=True, target_type='GaussianHeatmap') (top_down_eval.py:404-411, 455-463)."""
    heatmaps = heatmaps.copy()
    N, K, H, W = heatmaps.shape
    hm_preds, maxvals = get_max_preds(heatmaps)
    hm_preds = post_dark_udp(hm_preds, heatmaps, kernel)
    preds = hm_preds.copy()
    for i in range(N):
        preds[i] = transform_preds(preds[i], center[i], scale[i], [W, H], use_udp=True)
    return hm_preds, preds, maxvals


def fliplr_joints(joints_3d, joints_3d_visible, img_width, flip_pairs):
    """datasets/data_pipeline/RandomFlip.py:64-100."""
    jf, vf = joints_3d.copy(), joints_3d_visible.copy()
    for left, right in flip_pairs:
        jf[left, :] = joints_3d[right, :]
        jf[right, :] = joints_3d[left, :]
        vf[left, :] = joints_3d_visible[right, :]
        vf[right, :] = joints_3d_visible[left, :]
    jf[:, 0] = img_width - 1 - jf[:, 0]
    return jf * vf, vf


def random_flip(img, joints_3d, joints_3d_visible, center, flip_pairs):
    """TopDownRandomFlip.__call__ (RandomFlip.py:28-61) with the coin already tossed to 'flip'."""
    img = img[:, ::-1, :]
    j, v = fliplr_joints(joints_3d, joints_3d_visible, img.shape[1], flip_pairs)
    c = center.copy()
    c[0] = img.shape[1] - c[0] - 1
    return img, j, v, c


def candidate_bbox(center_maps, size_maps, num_candidates, image_size):
    """utils/HeatmapParser.py:52-85 (size_maps already region-averaged): top-k peaks, descending, ties by lower index."""
    b, h, w = center_maps.shape
    flat = center_maps.reshape(b, -1)
    idx = np.argsort(-flat, axis=1, kind="stable")[:, :num_candidates]
    val = np.take_along_axis(flat, idx, 1)
    cand = np.zeros((b, num_candidates, 5), np.float32)
    cand[..., 0] = idx % w
    cand[..., 1] = idx // w
    if size_maps is not None:
        sm = size_maps.reshape(b, 2, -1)
        cand[..., 2] = np.take_along_axis(sm[:, 0], idx, 1)
        cand[..., 3] = np.take_along_axis(sm[:, 1], idx, 1)
    cand[..., 2:4] = cand[..., 2:4].clip(0, 0.99)
    cand[..., 4] = val
    cand[..., :2] *= np.float32(image_size / w)
    cand[..., 2:4] *= np.float32(image_size)
    return cand


# ---------------------------------------------------------------------------------------------------------------------
# Training-time augmentation (datasets/build_dataset.py:111-122).  PARITY UNPINNED for the colour conversion: the reference
# calls cv2.cvtColor, cv2 is not importable here; the 8-bit conversions below restate OpenCV's documented algorithm
# (imgproc color_hsv: fixed-point RGB2HSV_b with hsv_shift = 12, hue range 180; HSV2RGB_b through the float sector formula).
# The reference's own integer arithmetic on the HSV planes (random_hsv.py:22-31) and its random draws are restated exactly.
def hsv_gains(n, hue_delta=5, saturation_delta=30, value_delta=30):
    """random_hsv.py:22-29 for n samples in sequence (numpy's GLOBAL generator, same calls in the same order as the
    reference's per-sample __call__): int16 [n, 3]."""
    out = np.zeros((n, 3), np.int16)
    for i in range(n):
        g = np.random.uniform(-1, 1, 3) * [hue_delta, saturation_delta, value_delta]
        g *= np.random.randint(0, 2, 3)
        out[i] = g.astype(np.int16)
    return out


def _cv_round(x):
    return np.rint(x)          # cvRound: round half to even


def bgr2hsv_u8(img):
    """cv2.cvtColor(img, cv2.COLOR_BGR2HSV) for uint8 [..., 3] (H in [0, 180))."""
    b, g, r = (img[..., k].astype(np.int32) for k in range(3))
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    idx = np.arange(256, dtype=np.float64)
    with np.errstate(divide="ignore"):
        sdiv = np.where(idx > 0, _cv_round((255 << 12) / np.maximum(idx, 1)), 0).astype(np.int64)
        hdiv = np.where(idx > 0, _cv_round((180 << 12) / (6.0 * np.maximum(idx, 1))), 0).astype(np.int64)
    s = (diff * sdiv[v] + (1 << 11)) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff)).astype(np.int64)
    h = (h * hdiv[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s, v], -1).astype(np.uint8)


def hsv2bgr_u8(hsv):
    """cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR) for uint8 [..., 3]: float sector formula, h * 6/180, s, v / 255, x 255, rounded."""
    h = hsv[..., 0].astype(np.float32) * np.float32(6.0 / 180.0)
    s = hsv[..., 1].astype(np.float32) * np.float32(1.0 / 255.0)
    v = hsv[..., 2].astype(np.float32) * np.float32(1.0 / 255.0)
    h = np.where(h >= 6, h - np.float32(6.0), h)
    sector = np.floor(h).astype(np.int32)
    f = (h - sector.astype(np.float32)).astype(np.float32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    f = np.where(bad, np.float32(0), f)
    one = np.float32(1.0)
    tab = np.stack([v, (v * (one - s)).astype(np.float32), (v * (one - (s * f).astype(np.float32))).astype(np.float32),
                    (v * (one - (s * (one - f)).astype(np.float32))).astype(np.float32)], -1)
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    pick = sd[sector]                                                   # [..., 3] indices of (b, g, r)
    bgr = np.take_along_axis(tab, pick, -1)
    bgr = np.where((hsv[..., 1] == 0)[..., None], v[..., None], bgr)
    return np.clip(_cv_round(bgr.astype(np.float32) * np.float32(255.0)), 0, 255).astype(np.uint8)


def hsv_jitter(img_bgr_u8, gains):
    """HSVRandomAug.__call__ (random_hsv.py:20-34) with the gains given: [H, W, 3] uint8 BGR -> jittered BGR."""
    hsv = bgr2hsv_u8(img_bgr_u8).astype(np.int16)
    g = np.asarray(gains, np.int16)
    hsv[..., 0] = (hsv[..., 0] + g[0]) % 180
    hsv[..., 1] = np.clip(hsv[..., 1] + g[1], 0, 255)
    hsv[..., 2] = np.clip(hsv[..., 2] + g[2], 0, 255)
    return hsv2bgr_u8(hsv.astype(np.uint8))


def random_scale_rotation(scale, rot_factor=40, scale_factor=0.5, rot_prob=0.6):
    """TopDownGetRandomScaleRotation.__call__ (topdown_affine.py:28-45) for a batch in sequence (numpy's global generator):
    scale [n, 2] -> (scale * s_factor, rotation [n])."""
    scale = np.asarray(scale)
    out_s, out_r = [], []
    for i in range(scale.shape[0]):
        sf, rf = scale_factor, rot_factor
        s_factor = np.clip(np.random.randn() * sf + 1, 1 - sf, 1 + sf)
        r_factor = np.clip(np.random.randn() * rf, -rf * 2, rf * 2)
        r = r_factor if np.random.rand() <= rot_prob else 0
        out_s.append(scale[i] * s_factor)
        out_r.append(r)
    return np.stack(out_s), np.array(out_r, np.float64)

"""HIP-backed mirror of loss/loss.py:69-114 (TopdownHeatmapLoss) and loss/heatmapLoss.py:228-265
(DistanceLoss, L2 / mean / class-balanced).  One pass over (output, target) forward, one backward."""
import ctypes as C

import torch
from torch import nn

from . import _lib


class DeviceScalar:
    """A loss value that stays on the GPU.  The reference calls `.item()` every step (loss.py:113), a host
    sync on the hot loop; this object converts lazily (float(), format, +)."""

    def __init__(self, t):
        self.t = t.detach()

    def item(self):
        return float(self.t.item())

    __float__ = item

    def __add__(self, o):
        return float(self) + float(o)

    __radd__ = __add__

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return f"DeviceScalar({float(self):.6g})"


class _BalancedMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target, weight, loss_weight, balance):
        _lib.require_device(output)
        o, t, weight = _lib.f32c(output), _lib.f32c(target), _lib.f32c(weight)
        if o.dim() == 5 and t.dim() == 4:
            # stacked hourglass output [N, S, K, H, W] (hourglassnet.py:136) against one target per sample: every stack is
            # supervised by the same target and weights (intermediate supervision).  The reference's own broadcast of a 4-D
            # target against the 5-D output is ill-formed (it needs N == S), see DESIGN.md.
            S = o.shape[1]
            t = t.unsqueeze(1).expand(-1, S, -1, -1, -1).contiguous()
            weight = weight.reshape(o.shape[0], 1, o.shape[2]).expand(-1, S, -1).contiguous()
        if t.shape != o.shape:
            raise _lib.LhnError(f"target shape {tuple(t.shape)} != output shape {tuple(o.shape)}")
        H, W = o.shape[-2:]
        N, K = o.numel() // (o.shape[-3] * H * W), o.shape[-3]      # leading dims fold into the sample count
        w = weight.reshape(N, K)
        acc = torch.empty(68, dtype=torch.float64, device=o.device)
        loss = torch.empty(1, dtype=torch.float32, device=o.device)
        L = _lib.lib()
        _lib.check(L.lhn_loss_balanced_mse_fwd(_lib.ptr(o), _lib.ptr(t), _lib.ptr(w), _lib.ptr(acc), _lib.ptr(loss),
                                               C.c_int64(N * K), C.c_int64(H * W), C.c_float(loss_weight),
                                               C.c_int(1 if balance else 0), _lib.stream()), "lhn_loss_balanced_mse_fwd")
        ctx.save_for_backward(o, t, w, acc)
        ctx.lw, ctx.balance = float(loss_weight), bool(balance)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        o, t, w, acc = ctx.saved_tensors
        H, W = o.shape[-2:]
        N, K = w.shape
        g = torch.empty_like(o)
        dl = _lib.f32c(dloss.reshape(1))
        L = _lib.lib()
        _lib.check(L.lhn_loss_balanced_mse_bwd(_lib.ptr(o), _lib.ptr(t), _lib.ptr(w), _lib.ptr(acc), _lib.ptr(dl),
                                               _lib.ptr(g), C.c_int64(N * K), C.c_int64(H * W), C.c_float(ctx.lw),
                                               C.c_int(1 if ctx.balance else 0), _lib.stream()), "lhn_loss_balanced_mse_bwd")
        return g, None, None, None, None


class DistanceLoss(nn.Module):
    """heatmapLoss.py:228-265 with loss_type 'L2' and reduction 'mean' (the forms the hot path uses)."""

    def __init__(self, loss_type="L2", reduction="mean", balance=True, value=0.5):
        super().__init__()
        if loss_type.lower() != "l2" or reduction != "mean" or value != 0.5:
            raise _lib.LhnError("DistanceLoss: only loss_type='L2', reduction='mean', value=0.5 are built")
        self.balance = balance

    def forward(self, output, target, target_weight, loss_weight=1.0):
        return _BalancedMSE.apply(output, target, target_weight, loss_weight, self.balance)


class _SimDRSmoothL1(torch.autograd.Function):
    """KLDiscretLoss (centernet_simdr_loss.py:6-39) on the decoded vectors: lhn_simdr_loss_fwd / _bwd."""

    @staticmethod
    def forward(ctx, px, py, tx, ty, w):
        N, K, Wd = px.shape
        Hd = py.shape[2]
        px, py, tx, ty, w = (_lib.f32c(t) for t in (px, py, tx, ty, w.reshape(N, K)))
        sums = torch.empty(3 * K, dtype=torch.float64, device=px.device)
        loss = torch.empty(1, dtype=torch.float32, device=px.device)
        _lib.check(_lib.lib().lhn_simdr_loss_fwd(_lib.ptr(px), _lib.ptr(py), _lib.ptr(tx), _lib.ptr(ty), _lib.ptr(w),
                                                 _lib.ptr(sums), _lib.ptr(loss), N, K, Wd, Hd, _lib.stream()),
                   "lhn_simdr_loss_fwd")
        ctx.save_for_backward(px, py, tx, ty, sums)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        px, py, tx, ty, sums = ctx.saved_tensors
        N, K, Wd = px.shape
        dpx, dpy = torch.empty_like(px), torch.empty_like(py)
        g = _lib.f32c(gout.reshape(1))
        _lib.check(_lib.lib().lhn_simdr_loss_bwd(_lib.ptr(px), _lib.ptr(py), _lib.ptr(tx), _lib.ptr(ty), _lib.ptr(sums),
                                                 _lib.ptr(g), _lib.ptr(dpx), _lib.ptr(dpy), N, K, Wd, py.shape[2],
                                                 _lib.stream()), "lhn_simdr_loss_bwd")
        return dpx, dpy, None, None, None


class SimDRLoss(nn.Module):
    """centernet_simdr_loss.py:42-71: two shared Linear decoders heatmap[H*W] -> 1-D x / y vectors (plain library GEMMs:
    torch -> rocBLAS) and the SmoothL1 loss above.  Same attribute names / state_dict keys as the reference."""

    def __init__(self, cfg=None):
        super().__init__()
        k = cfg.PIPELINE.simdr_split_ratio
        self.simdr_width = int(k * cfg.DATASET.image_size[0])
        self.simdr_height = int(k * cfg.DATASET.image_size[1])
        in_features = int(cfg.DATASET.heatmap_size[0] * cfg.DATASET.heatmap_size[1])
        self.x_shared_decoder = nn.Linear(in_features, self.simdr_width)
        self.y_shared_decoder = nn.Linear(in_features, self.simdr_height)

    def decode(self, heatmap):
        flat = heatmap.flatten(start_dim=2)
        return self.x_shared_decoder(flat), self.y_shared_decoder(flat)

    def forward(self, heatmap, simdr_x, simdr_y, target_weight):
        px, py = self.decode(heatmap)
        return _SimDRSmoothL1.apply(px, py, simdr_x, simdr_y, target_weight)


class TopdownHeatmapLoss(nn.Module):
    """loss.py:69-114.  meta['target'] [N,K,H,W], meta['target_weight'] [N,K,1] may live on CPU or device."""

    def __init__(self, cfg):
        super().__init__()
        self.heatmap_loss = DistanceLoss(loss_type=cfg.LOSS.get("dl_type", "L2"), reduction="mean",
                                         balance=cfg.MODEL.name != "atthandnet")
        self.simdr_loss = SimDRLoss(cfg) if cfg.PIPELINE.simdr_split_ratio > 0 else None      # loss.py:81-85
        self.loss_weight = cfg.LOSS.loss_weight
        if cfg.LOSS.auto_weight:
            raise _lib.LhnError("LOSS.auto_weight is outside the hot path")

    def forward(self, output, meta):
        device = output.device
        target = meta["target"].to(device, non_blocking=True)
        weight = meta["target_weight"].to(device, non_blocking=True)
        loss = self.heatmap_loss(output, target, weight, float(self.loss_weight[0]))
        loss_dict = {"heatmap": DeviceScalar(loss)}
        if self.simdr_loss is not None:                                                        # loss.py:102-106
            ls = float(self.loss_weight[1]) * self.simdr_loss(output, meta["simdr_x"].to(device, non_blocking=True),
                                                              meta["simdr_y"].to(device, non_blocking=True), weight)
            loss_dict["simdr"] = DeviceScalar(ls)
            loss = loss + ls
        return loss, loss_dict


topdownheatmaploss = TopdownHeatmapLoss


def get_loss(cfg):
    name = cfg.LOSS.type.lower()
    if name != "topdownheatmaploss":
        raise _lib.LhnError(f"loss <{cfg.LOSS.type}> is outside the hot path")
    return TopdownHeatmapLoss(cfg)

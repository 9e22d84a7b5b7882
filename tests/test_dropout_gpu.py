"""Dropout switched ON (what bench.py times: cfg.MODEL.ca_dropout = 0.3, batch 64 at 256x256).

The masks are drawn by the plan per attention module as [N, C] tensors of {0, 1/keep}; `Engine.mask_fn` injects given masks
instead, and `oracle.torch_ref.install_masks` makes the float64 oracle multiply by the same ones (common.py:40-66: the
reference's nn.Dropout2d zeroes whole (n, c) planes of the 1x1 pooled tensor and scales the rest by 1/(1-p))."""
import copy

import numpy as np
import pytest
import torch

from litehandnet_amd.config import litehandnet_cfg
from oracle import heatmap_np as onp
from oracle import synth, torch_ref

pytestmark = pytest.mark.gpu
P = 0.3


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _attach(ours, n, seed, p=P):
    """Engine with injected masks for `ours`; returns {module name: mask [n, C]} as drawn (lazily, per attention module)."""
    from litehandnet_amd.engine import Engine
    names = {id(m): k for k, m in ours.named_modules()}
    masks = {}

    def fill(plan):
        for mod, view in plan.mask_slices:
            k = names[id(mod)]
            if k not in masks:
                r = np.random.Generator(np.random.PCG64([seed, len(masks)]))
                masks[k] = torch.from_numpy(((r.random(tuple(view.shape)) < 1 - p) / (1 - p)).astype(np.float32))
            view.copy_(masks[k])
    eng = Engine(ours, p_drop=p)
    eng.mask_fn = fill
    ours.__dict__["_engine"] = eng
    return masks


def _block(ours, ref, x, dev, seed, grad_tol=1e-3, no_dx=False):
    """forward + backward of a block with shared dropout masks, against the float64 oracle; bars as in test_model_gpu._check_block."""
    sd = synth.synth_state_dict(ref, seed)
    ref.load_state_dict(sd); ours.load_state_dict(sd)
    ours.to(dev).train(); ref.train()
    masks = _attach(ours, x.shape[0], seed + 500)
    xg = x.clone().to(dev).requires_grad_(not no_dx)
    yg = ours(xg)
    assert masks, "no dropout mask was drawn"
    torch_ref.install_masks(ref, masks)
    ref32 = copy.deepcopy(ref)
    ref = ref.double()
    g = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 100)).standard_normal(tuple(yg.shape)).astype(np.float32))
    xr = x.double().clone().requires_grad_()
    yr = ref(xr)
    yr = yr[-1] if isinstance(yr, (tuple, list)) else yr
    yr.backward(g.double())
    x32 = x.clone().requires_grad_()
    y32 = ref32(x32)
    y32 = y32[-1] if isinstance(y32, (tuple, list)) else y32
    y32.backward(g)
    assert _rel(yg, yr) < max(1e-4, 3 * _rel(y32, yr)), ("forward", _rel(yg, yr), _rel(y32, yr))
    yg.backward(g.to(dev))
    if not no_dx:
        assert _rel(xg.grad, xr.grad) < max(grad_tol, 3 * _rel(x32.grad, xr.grad)), ("dx", _rel(xg.grad, xr.grad))
    rp, rp32 = dict(ref.named_parameters()), dict(ref32.named_parameters())
    floor = 1e-3 * max(float(v.grad.double().norm()) for v in rp.values())
    e32s = {k: float((rp32[k].grad.double() - rp[k].grad.double()).norm() / (rp[k].grad.double().norm() + floor)) for k in rp}
    for k, p in ours.named_parameters():
        e = float((p.grad.cpu().double() - rp[k].grad.double()).norm() / (rp[k].grad.double().norm() + floor))
        assert e < max(grad_tol, 3 * e32s[k], 1.5 * max(e32s.values())), (k, e, e32s[k])
    return masks


def test_masks_are_per_plane_and_scaled(dev):
    """Default (un-injected) masks of a training forward: one value per (n, c), 0 or exactly 1/(1-p), about p of them zero,
    different per attention module and per step; eval mode draws none."""
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg("B")
    assert cfg.MODEL.get("ca_dropout", 0.3) == 0.3
    m = get_model(cfg).to(dev).train()
    x = synth.synth_images(16, 64, 3).to(dev)
    with torch.no_grad():
        m(x)
    plan = next(iter(m.__dict__["_engine"].plans.values()))
    assert len(plan.mask_slices) == 8                       # 2 MSRB x 2 + 4 gated RepBasicUnits (stem, neck)
    first = [v.clone() for _, v in plan.mask_slices]
    keep = np.float32(1.0) / np.float32(1.0 - P)
    tot = zero = 0
    for v in first:
        assert v.shape == (16, 128)
        a = v.cpu().numpy()
        assert np.all((a == 0) | (a == np.float32(1.0 / (1.0 - P))) | (a == keep))
        tot += a.size
        zero += int((a == 0).sum())
    assert abs(zero / tot - P) < 5 * np.sqrt(P * (1 - P) / tot)
    assert not torch.equal(first[0], first[1])
    with torch.no_grad():
        m(x)
    assert not torch.equal(first[0], plan.mask_slices[0][1])
    snap = plan.mask_slices[0][1].clone()
    m.eval()
    with torch.no_grad():
        m(x)                                                 # eval plan: no masks at all / the training plan's are untouched
    assert torch.equal(snap, plan.mask_slices[0][1])


def test_channel_attention_with_dropout(dev):
    from litehandnet_amd.common import ChannelAttension
    r = np.random.Generator(np.random.PCG64(5))
    x = torch.from_numpy(r.standard_normal((8, 64, 12, 12)).astype(np.float32))
    masks = _block(ChannelAttension(64, p_drop=P), torch_ref.ChannelAttension(64, P), x, dev, seed=81)
    assert 0 < float((masks[""] == 0).float().mean()) < 1


def test_msrb_with_dropout(dev):
    from litehandnet_amd import litehourglass as lh
    r = np.random.Generator(np.random.PCG64(6))
    x = torch.from_numpy(r.standard_normal((8, 64, 16, 16)).astype(np.float32))
    _block(lh.MSRB(64, 64, "ca", p_drop=P), torch_ref.MSRB(64, 64, "ca", P), x, dev, seed=82)


@pytest.mark.parametrize("variant", ["B", "M"])
def test_model_with_dropout_small(dev, variant):
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg(variant)
    ours, ref = get_model(cfg), torch_ref.get_model(cfg, p_drop=P)
    # (mynet: 16 samples -- with 8, the BatchNorm over 8 dropped-out attention values leaves single gradients ill-conditioned
    # enough that a ReLU sign flip within one fp32 ulp moves them by several percent, see test_odd_batches)
    n = 16 if variant == "M" else 8
    _block(ours, ref, synth.synth_images(n, 64, 9), dev, seed=83, no_dx=True, grad_tol=3e-2)


@pytest.mark.parametrize("variant", ["B", "A"])
def test_bench_config_bs64_256(dev, variant):
    """BASELINE configs 2 and 3 at FULL size, exactly what bench.py times: batch 64, 256x256, train-mode BatchNorm, Dropout2d
    p = 0.3 (shared masks), forward + TopdownHeatmapLoss + backward.  Arbiter = the oracle in float64; yardstick = the same
    oracle in fp32 on the CPU (what the reference computes).  Heatmap <= max(1e-4, 3 x fp32 error); integer argmax
    coordinates equal to float64's except at near-ties within the measured error (no more of them than the fp32 CPU run has); loss and per-parameter gradient norms within 3 x the
    fp32 run's own error (floor 1e-3); PCK@0.2 of the decoded keypoints against the float64 decode = 1 within 0.1 %."""
    from litehandnet_amd import get_loss, get_model, heatmap
    n, size, seed = 64, 256, 7
    cfg = litehandnet_cfg(variant)
    ours, ref = get_model(cfg), torch_ref.get_model(cfg, p_drop=P)
    sd = synth.synth_state_dict(ref, seed)
    ref.load_state_dict(sd); ours.load_state_dict(sd)
    ours.to(dev).train(); ref.train()
    masks = _attach(ours, n, seed + 500)
    x = synth.synth_images(n, size, seed)
    j = synth.synth_joints(n, 21, size, seed + 1)
    tgt = torch.from_numpy(np.stack([onp.msra_generate_target(a, np.ones_like(a), [size, size], [64, 64])[0] for a in j]))
    tw = torch.ones(n, 21, 1)
    y = ours(x.to(dev))
    loss, _ = get_loss(cfg)(y, {"target": tgt, "target_weight": tw})
    loss.backward()
    torch_ref.install_masks(ref, masks)
    ref32 = copy.deepcopy(ref)
    y32 = ref32(x)
    l32 = cfg.LOSS.loss_weight[0] * torch_ref.distance_loss(y32, tgt, tw)
    l32.backward()
    g32 = {k: float(p.grad.norm()) for k, p in ref32.named_parameters()}
    y32 = y32.detach().numpy()
    del ref32
    ref = ref.double()
    y64 = ref(x.double())
    l64 = cfg.LOSS.loss_weight[0] * torch_ref.distance_loss(y64, tgt.double(), tw.double())
    l64.backward()
    y64n = y64.detach().numpy()
    scale = np.abs(y64n).max()
    e32 = np.abs(y32 - y64n).max() / scale
    err = np.abs(y.detach().cpu().numpy() - y64n).max() / scale
    assert err <= max(1e-4, 3 * e32), (err, e32)
    assert abs(float(loss.detach()) - float(l64)) <= max(3 * abs(float(l32) - float(l64)), 1e-5 * abs(float(l64)))
    g64 = {k: float(p.grad.norm()) for k, p in ref.named_parameters()}
    floor = 1e-3 * max(g64.values())
    worst = max(abs(float(p.grad.norm()) - g64[k]) / (g64[k] + floor) for k, p in ours.named_parameters())
    worst32 = max(abs(g32[k] - g64[k]) / (g64[k] + floor) for k in g64)
    from test_model_gpu import MODEL_GRAD_FACTOR, MODEL_GRAD_FLOOR
    assert worst <= max(MODEL_GRAD_FACTOR * worst32, MODEL_GRAD_FLOOR), (worst, worst32)              # see test_model_gpu._model_case
    p, _ = heatmap._get_max_preds(y.detach())
    p64, _ = onp.get_max_preds(y64n.astype(np.float32))
    p32, _ = onp.get_max_preds(y32)
    same32 = (p32 == p64).all(-1)
    pn = p.cpu().numpy()
    # Integer argmax coordinates: bit-exact against float64 wherever the map has a UNIQUE maximum at fp32 resolution.  Our map
    # is the float64 map perturbed by at most `err * scale`, so a different argmax is only legitimate at a near-tie: the
    # float64 value at the position we picked must lie within twice that perturbation of the float64 maximum.  Anything
    # else is a real decode error.  (64 x 21 = 1344 key points per batch; near-ties are counted and bounded.)
    diff = ~(pn == p64).all(-1)
    flat = y64n.reshape(n, 21, -1)
    ours_idx = (pn[..., 1] * 64 + pn[..., 0]).astype(np.int64).clip(0)
    gap = flat.max(-1) - np.take_along_axis(flat, ours_idx[..., None], -1)[..., 0]
    assert (gap[diff] <= 2 * max(err, 1e-6) * scale).all(), (gap[diff].max(), err * scale)
    assert diff.sum() <= max(2, int((~same32).sum()) + 2), (int(diff.sum()), int((~same32).sum()))   # no worse than the fp32 CPU run
    # PCK@0.2 (top_down_eval.py:129-165) of our decode against the float64 decode, normalised by the 64x64 map
    acc, avg, cnt = onp.keypoint_pck_accuracy(pn, p64, np.ones((n, 21), bool), 0.2, np.full((n, 2), 64.0, np.float32))
    assert avg >= 0.999, avg
    from conftest import parity_record
    parity_record(f"bench_config_bs64_256_{variant}", heatmap_err=err, heatmap_err_cpu_fp32=e32, grad_norm_worst=worst, grad_norm_cpu_fp32_worst=worst32,
                  grad_norm_bar=max(MODEL_GRAD_FACTOR * worst32, MODEL_GRAD_FLOOR), argmax_disagree_vs_f64=int(diff.sum()),
                  argmax_disagree_vs_cpu_fp32=int((~(pn == p32).all(-1)).sum()), argmax_cpu_fp32_disagree_vs_f64=int((~same32).sum()),
                  keypoints=int(n * 21), pck_vs_f64_decode=float(avg), pck_delta=float(1.0 - avg))
    print(f"[{variant} bs64 256 p=0.3] heatmap err vs f64: hip {err:.2e} / cpu-fp32 {e32:.2e}; grad-norm: hip {worst:.2e} / "
          f"cpu-fp32 {worst32:.2e}; argmax agree {float((pn == p64).all(-1).mean()):.4f} (fp32 cpu {float(same32.mean()):.4f}), near-ties {int(diff.sum())}; PCK {avg:.4f}")

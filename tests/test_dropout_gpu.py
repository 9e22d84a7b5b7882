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


def test_mask_injector_is_read_at_run_time_and_no_grad_forward_between(dev):
    """`Engine.mask_fn` set AFTER the first forward of a shape takes effect, and clearing it goes back to random masks (the
    plan used to copy it once, at creation).  And the guard of Engine against stale forwards: a training forward, then a
    torch.no_grad() forward of the same shape (its own plan and workspace), then the backward of the first one -- allowed,
    and equal to the backward without the extra forward; a second grad-enabled forward in between is refused."""
    from litehandnet_amd import _lib, get_model
    cfg = litehandnet_cfg("B")
    m = get_model(cfg).to(dev).train()
    x = synth.synth_images(4, 64, 3).to(dev)
    g = torch.ones(4, 21, 16, 16, device=dev)
    y0 = m(x)                                                 # random masks, creates the training plan
    eng = m.__dict__["_engine"]
    plan = [p for k, p in eng.plans.items() if k[1]][0]
    ones = lambda pl: [v.fill_(1.0) for _, v in pl.mask_slices]      # noqa: E731
    eng.mask_fn = ones
    y1 = m(x)
    assert all(float(v.min()) == 1.0 == float(v.max()) for _, v in plan.mask_slices)
    m.zero_grad()
    y1.backward(g)
    g1 = torch.cat([p.grad.flatten().clone() for p in m.parameters()])
    # the same step with a no_grad forward of the same shape in between
    rs = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(rs)
    y2 = m(x)
    with torch.no_grad():
        m(x)
    m.zero_grad()
    y2.backward(g)
    g2 = torch.cat([p.grad.flatten() for p in m.parameters()])
    assert float((g1 - g2).norm()) <= 1e-3 * float(g1.norm())      # (running statistics moved once more: same masks, same batch)
    y3 = m(x)
    m(x)                                                       # a second grad-enabled forward of the same shape ...
    with pytest.raises(_lib.LhnError):
        y3.backward(g)                                         # ... makes the first one stale
    eng.mask_fn = None
    m(x)
    assert any(float(v.min()) == 0.0 for _, v in plan.mask_slices)            # random masks again (p = 0.3 over 8 x 4 x 128 values)
    del y0


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

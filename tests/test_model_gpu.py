"""GPU parity: HIP-backed blocks and the full MSRB hourglass (variant B) against the torch-CPU oracle on the
same seeded inputs and weights, forward and backward, plus the committed golden vectors."""
import os

import numpy as np
import pytest
import torch

from litehandnet_amd.config import litehandnet_cfg
from oracle import heatmap_np as onp
from oracle import synth, torch_ref

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-4     # fp32: max-abs error relative to the tensor's max-abs (different summation order, fused BN)
GRAD_TOL = 2e-3    # per-parameter gradient, relative to that gradient's norm


def _no_dropout(m):
    for x in m.modules():
        if isinstance(x, torch.nn.Dropout2d):
            x.p = 0.0


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _check_block(ours, ref, x, dev, seed=0, fwd_tol=FWD_TOL, grad_tol=GRAD_TOL, pick=None):
    sd = synth.synth_state_dict(ref, seed)
    ref.load_state_dict(sd)
    ours.load_state_dict(sd)
    ours.to(dev)
    ref.train(); ours.train()
    _no_dropout(ref)
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    if pick is not None:
        yr = yr[pick]
    g = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 100)).standard_normal(tuple(yr.shape)).astype(np.float32))
    yr.backward(g)
    xg = x.clone().to(dev).requires_grad_()
    yg = ours(xg)
    assert yg.shape == yr.shape
    assert _rel(yg, yr) < fwd_tol, ("forward", _rel(yg, yr))
    yg.backward(g.to(dev))
    assert _rel(xg.grad, xr.grad) < grad_tol, ("dx", _rel(xg.grad, xr.grad))
    rp = dict(ref.named_parameters())
    for k, p in ours.named_parameters():
        assert p.grad is not None, k
        e = float((p.grad.cpu().double() - rp[k].grad.double()).norm() / (rp[k].grad.double().norm() + 1e-12))
        assert e < grad_tol, (k, e)
    # running statistics (momentum 0.1, unbiased variance) after one training step
    for k, v in ours.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.allclose(v.cpu(), ref.state_dict()[k], rtol=1e-4, atol=1e-5), k
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref.state_dict()[k]), k


def _x(n, c, h, w, seed=0):
    r = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(r.standard_normal((n, c, h, w)).astype(np.float32))


@pytest.mark.parametrize("cin,cout,act,inplace", [(64, 64, "lrelu", False), (32, 128, "lrelu", True), (128, 32, None, False)])
def test_repconv_pointwise(dev, cin, cout, act, inplace):
    from litehandnet_amd import repblocks
    a = torch.nn.LeakyReLU if act else None
    _check_block(repblocks.RepConv(cin, cout, 1, activation=a, inplace=inplace),
                 torch_ref.RepConv(cin, cout, 1, activation=a, inplace=inplace), _x(3, cin, 12, 20), dev)


@pytest.mark.parametrize("c,dil,stride", [(32, 1, 1), (64, 2, 1), (32, 1, 2)])
def test_repconv_depthwise(dev, c, dil, stride):
    from litehandnet_amd import repblocks
    _check_block(repblocks.RepConv(c, c, 3, stride, dil, dil, groups=c, activation=None),
                 torch_ref.RepConv(c, c, 3, stride, dil, dil, groups=c, activation=None), _x(2, c, 16, 24), dev)


@pytest.mark.parametrize("ca", ["none", "ca"])
def test_msrb(dev, ca):
    from litehandnet_amd import litehourglass as lh
    _check_block(lh.MSRB(64, 64, ca, p_drop=0.0), torch_ref.MSRB(64, 64, ca, 0.0), _x(4, 64, 16, 16), dev)


@pytest.mark.parametrize("ca", ["none", "ca"])
def test_rep_basic_unit(dev, ca):
    from litehandnet_amd import litehourglass as lh
    _check_block(lh.RepBasicUnit(64, 64, ca, p_drop=0.0), torch_ref.RepBasicUnit(64, 64, ca, 0.0), _x(4, 64, 12, 12), dev)


def test_hourglass_B_block(dev):
    from litehandnet_amd import litehourglass as lh
    _check_block(lh.EncoderDecoder(4, 32, "ca", "none", p_drop=0.0), torch_ref._HourglassB(4, 32, "ca", "none", 0.0),
                 _x(2, 32, 32, 32), dev, grad_tol=5e-3, pick=-1)


def _model_case(dev, golden_dir, tag, **kw):
    from litehandnet_amd import get_loss, get_model
    g = np.load(os.path.join(golden_dir, f"model_{tag}.npz"))
    cfg = litehandnet_cfg("B", **kw)
    cfg.MODEL["ca_dropout"] = 0.0
    n, size, seed = int(g["n"]), int(g["size"]), int(g["seed"])
    m = get_model(cfg)
    m.load_state_dict(synth.synth_state_dict(m, seed))
    m.to(dev).train()
    x = synth.synth_images(n, size, seed).to(dev)
    y = m(x)
    err = np.abs(y.detach().cpu().numpy() - g["heatmap"]).max() / np.abs(g["heatmap"]).max()
    assert err < FWD_TOL, err
    hs = size // 4
    j = synth.synth_joints(n, 21, size, seed + 1)
    tgt = np.stack([onp.msra_generate_target(a, np.ones_like(a), [size, size], [hs, hs])[0] for a in j])
    meta = {"target": torch.from_numpy(tgt), "target_weight": torch.from_numpy(g["target_weight"])}
    loss, _ = get_loss(cfg)(y, meta)
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    loss.backward()
    gn = dict(zip(g["grad_keys"].tolist(), g["grad_norms"].tolist()))
    worst = 0.0
    for k, p in m.named_parameters():
        e = abs(float(p.grad.norm()) - gn[k]) / (gn[k] + 1e-9)
        worst = max(worst, e)
        assert e < 5e-3, (k, e, gn[k])
    bk = str(g["bn_key"])
    assert np.allclose(m.state_dict()[bk].cpu().numpy(), g["bn_running_mean"], rtol=1e-4, atol=1e-6)
    # integer argmax of the produced heatmap vs the reference's heatmap: bit-exact coordinates
    from litehandnet_amd import heatmap
    p, _ = heatmap._get_max_preds(y.detach())
    pref, _ = onp.get_max_preds(g["heatmap"])
    agree = (p.cpu().numpy() == pref).all(-1).mean()
    assert agree >= 0.999, agree


def test_model_B_64_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "B_64")


def test_model_B_256_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "B_256")


def test_model_Bca_64_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "Bca_64", rbu_ca="ca")


def test_model_B_eval_golden(dev, golden_dir):
    from litehandnet_amd import get_model
    g = np.load(os.path.join(golden_dir, "model_B_64_eval.npz"))
    m = get_model(litehandnet_cfg("B"))
    m.load_state_dict(synth.synth_state_dict(m, int(g["weights_seed"])))
    m.to(dev).eval()
    with torch.no_grad():
        y = m(synth.synth_images(2, 64, int(g["seed"])).to(dev))
    assert np.abs(y.cpu().numpy() - g["heatmap"]).max() <= FWD_TOL * np.abs(g["heatmap"]).max()


def test_state_dict_contract(dev):
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg("B")
    ours, ref = get_model(cfg), torch_ref.get_model(cfg)
    assert list(ours.state_dict()) == list(ref.state_dict())
    for (k, a), (_, b) in zip(ours.state_dict().items(), ref.state_dict().items()):
        assert a.shape == b.shape, k

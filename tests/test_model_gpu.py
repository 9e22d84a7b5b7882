"""GPU parity: HIP-backed blocks and the full MSRB hourglass (variant B) against the torch-CPU oracle on the
same seeded inputs and weights, forward and backward, plus the committed golden vectors."""
import copy
import os

import numpy as np
import pytest
import torch

from litehandnet_amd.config import litehandnet_cfg
from oracle import heatmap_np as onp
from oracle import synth, torch_ref
from conftest import parity_record

pytestmark = pytest.mark.gpu
STRICT = os.environ.get("LHN_STRICT_BARS") == "1"
# whole-network gradient NORMS: floor 1e-3 of the largest norm; factor 3 x the reference's own fp32 error in the strict
# (deterministic) run, 4 x in the default mode where the arrival order of the atomics moves the worst parameter run to run
MODEL_GRAD_FACTOR = 3.0 if STRICT else float(os.environ.get("LHN_MODEL_GRAD_FACTOR", "3.0"))
MODEL_GRAD_FLOOR = 1e-3
WHOLE_MODEL_GRAD_TOL = 2e-2      # elementwise gradients of whole small networks (chaotic ReLU flips, see test_width_256)

FWD_TOL = 1e-4     # fp32 vs the float64 oracle: max-abs error relative to the tensor's max-abs
GRAD_TOL = 1e-3    # gradients vs the float64 oracle, relative to the gradient's norm (floored, see _check_block)


def _no_dropout(m):
    for x in m.modules():
        if isinstance(x, (torch.nn.Dropout2d, torch.nn.Dropout)):
            x.p = 0.0


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _check_block(ours, ref, x, dev, seed=0, fwd_tol=FWD_TOL, grad_tol=GRAD_TOL, pick=None, no_dx=False, whole=None):
    """HIP block vs the oracle evaluated in float64 (the same restatement, run in double, is the arbiter:
    some of these blocks are ill-conditioned enough that torch's own fp32 CPU result is off by >1e-2)."""
    sd = synth.synth_state_dict(ref, seed)
    ref.load_state_dict(sd)
    ours.load_state_dict(sd)
    ours.to(dev)
    ref.train(); ours.train()
    _no_dropout(ref)
    ref32 = copy.deepcopy(ref)
    ref = ref.double()
    xr = x.double().clone().requires_grad_()
    yr = ref(xr)
    if pick is not None:
        yr = yr[pick]
    g = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 100)).standard_normal(tuple(yr.shape)).astype(np.float32))
    yr.backward(g.double())
    # the same block in fp32 on the CPU (= what the reference computes): its distance from the float64 result
    # is the conditioning yardstick -- the HIP path may not be more than 3x further away than that
    x32 = x.clone().requires_grad_()
    y32 = ref32(x32)
    if pick is not None:
        y32 = y32[pick]
    y32.backward(g)
    xg = x.clone().to(dev).requires_grad_(not no_dx)      # whole models take the image without a gradient
    yg = ours(xg)
    assert yg.shape == yr.shape
    assert _rel(yg, yr) < max(fwd_tol, 3 * _rel(y32, yr)), ("forward", _rel(yg, yr), _rel(y32, yr))
    yg.backward(g.to(dev))
    if not no_dx:
        assert _rel(xg.grad, xr.grad) < max(grad_tol, 3 * _rel(x32.grad, xr.grad)), ("dx", _rel(xg.grad, xr.grad), _rel(x32.grad, xr.grad))
    rp, rp32 = dict(ref.named_parameters()), dict(ref32.named_parameters())
    # some gradients are mathematically zero (a BN shift feeding a conv that is itself batch-normalised):
    # measure every error against max(own norm, 1e-3 x the largest gradient norm of the block)
    floor = 1e-3 * max(float(v.grad.double().norm()) for v in rp.values())
    # bar per parameter: 3x the reference's own fp32 error on that parameter, or the reference's WORST fp32 error in the
    # block (weight-gradient atomics make our low bits vary run to run; a parameter sitting exactly at 3x must not flake)
    e32s = {k: float((rp32[k].grad.double() - rp[k].grad.double()).norm() / (rp[k].grad.double().norm() + floor)) for k in rp}
    worst32 = max(e32s.values())
    # WHOLE networks (no_dx: the block reads the image): gradients are chaotic at this level -- a max-pool tie or a ReLU
    # derivative within one fp32 ulp of zero routes differently under another summation order and moves one parameter by
    # percents, in torch's own fp32 run as much as in ours -- so besides 3x the reference's error on the SAME parameter a
    # parameter may be as far off as twice the reference's WORST parameter; both modes (a deterministic order is still a
    # different order than torch's).  Blocks: 1.5x the worst in the default mode (atomics), nothing in the strict mode.
    whole = no_dx if whole is None else whole
    esc = 2.0 * worst32 if whole else (0.0 if STRICT else 1.5 * worst32)
    worst_e, worst_ratio, worst_k, fails, flips = 0.0, 0.0, "", [], []
    for k, p in ours.named_parameters():
        assert p.grad is not None, k
        den = rp[k].grad.double().norm() + floor
        e = float((p.grad.cpu().double() - rp[k].grad.double()).norm() / den)
        if e / max(grad_tol, 3 * e32s[k]) > worst_ratio:
            worst_e, worst_ratio, worst_k = e, e / max(grad_tol, 3 * e32s[k]), k
        if not e < max(grad_tol, 3 * e32s[k], esc):
            # ONE flipped ReLU (an activation within an ulp of zero in the last combine of a network, where nothing averages it out)
            # moves exactly one output channel of the layers that produce it: the BatchNorm shift and the weight row of that channel.
            # A whole-network parameter that is within its bar once its single worst leading-dim row is left out is recorded as such;
            # at most 4 per network, all naming the same channel (measured: Lite-HRNet-18 at N = 2 in the deterministic mode,
            # channel 5 of stage2.2.fuse_layers.1.*: one d(y) element of 0.54 against the float64 oracle).
            d = (p.grad.cpu().double() - rp[k].grad.double()).reshape(p.shape[0], -1)
            row = int(d.norm(dim=1).argmax())
            d[row] = 0
            if whole and p.shape[0] > 1 and float(d.norm() / den) < max(grad_tol, 3 * e32s[k], esc):
                flips.append((k, row, e, float(d.norm() / den)))
            else:
                fails.append((k, e, e32s[k], worst32))
    if flips and (len(flips) > 4 or len({r for _, r, _, _ in flips}) > 1):
        fails += flips
    test = os.environ.get("PYTEST_CURRENT_TEST", "block").split("::")[-1].split(" ")[0]
    parity_record(f"{test}/{type(ref).__name__}/seed{seed}", fwd_err=_rel(yg, yr), fwd_err_ref_fp32=_rel(y32, yr), fwd_bar=max(fwd_tol, 3 * _rel(y32, yr)),
                  grad_worst_err=worst_e, grad_worst_over_own_bar=worst_ratio, grad_worst_param=worst_k, grad_worst_param_ref_fp32_err=e32s.get(worst_k, 0.0),
                  grad_ref_fp32_worst=worst32, grad_tol=grad_tol, whole_network=bool(whole), failed=len(fails),
                  single_channel_exceptions=[f"{k}[{r}]: {e:.3e} -> {e2:.3e}" for k, r, e, e2 in flips])
    assert not fails, fails[:4]
    # running statistics (momentum 0.1, unbiased variance) after one training step
    for k, v in ours.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.allclose(v.cpu().double(), ref.state_dict()[k], rtol=1e-4, atol=1e-5), k
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref32.state_dict()[k]), k


def _x(n, c, h, w, seed=0):
    r = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(r.standard_normal((n, c, h, w)).astype(np.float32))


@pytest.mark.parametrize("cin,cout,act,inplace", [(64, 64, "lrelu", False), (32, 128, "lrelu", True), (128, 32, None, False)])
def test_repconv_pointwise(dev, cin, cout, act, inplace):
    from litehandnet_amd import repblocks
    a = torch.nn.LeakyReLU if act else None
    _check_block(repblocks.RepConv(cin, cout, 1, activation=a, inplace=inplace),
                 torch_ref.RepConv(cin, cout, 1, activation=a, inplace=inplace), _x(3, cin, 12, 20), dev)


def test_silu_activation(dev):
    """activation='silu' (liteHandNet.py:203-205 maps the cfg string to nn.SiLU): RepConv stores
    silu(BN(conv)) through the elementwise combine, RepBlock / BasicBlock / BottleNeck apply it in their own combine."""
    from litehandnet_amd import get_model, liteHandNet, repblocks
    S = torch.nn.SiLU
    _check_block(repblocks.RepConv(64, 64, 1, activation=S), torch_ref.RepConv(64, 64, 1, activation=S), _x(3, 64, 12, 20), dev)
    _check_block(repblocks.RepConv(32, 32, 3, 1, 1, groups=32, activation=S, inplace=True),
                 torch_ref.RepConv(32, 32, 3, 1, 1, groups=32, activation=S, inplace=True), _x(2, 32, 16, 24), dev, seed=1)
    _check_block(repblocks.RepBlock(32, 32, 7, 1, 3, groups=32, activation=S),
                 torch_ref.RepBlock(32, 32, 7, 1, 3, groups=32, activation=S), _x(2, 32, 16, 16), dev, seed=2)
    _check_block(liteHandNet.BottleNeck(128, 4, S), torch_ref.BottleNeck(128, 4, S), _x(4, 128, 16, 16), dev, seed=3)
    cfg = litehandnet_cfg("A", activation="silu")
    cfg.MODEL["ca_dropout"] = 0.0
    ours, ref = get_model(cfg), torch_ref.get_model(cfg, p_drop=0.0)
    _check_block(ours, ref, synth.synth_images(8, 64, 13), dev, seed=14, no_dx=True, grad_tol=WHOLE_MODEL_GRAD_TOL)      # whole model: see _model_case


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
    _check_block(lh.EncoderDecoder(4, 64, "ca", "none", p_drop=0.0), torch_ref._HourglassB(4, 64, "ca", "none", 0.0),
                 _x(8, 64, 64, 64), dev, pick=-1)


def test_stem_B(dev):
    from litehandnet_amd import litehourglass as lh
    _check_block(lh.Stem(128, p_drop=0.0), torch_ref._StemB(128, 0.0), _x(2, 3, 64, 64), dev, no_dx=True)


def _model_case(dev, golden_dir, tag, variant="B", **kw):
    """Full variant-B model, forward + TopdownHeatmapLoss + backward.

    Arbiter = the oracle in float64.  Bar: the HIP fp32 result must be at least as close to exact arithmetic
    as the REFERENCE's own fp32 CPU run (the committed golden vector) is, up to a factor 3 (floors 1e-4 for
    the heatmap, 1e-3 for gradient norms; these N=2 fixtures are ill-conditioned: the reference's own fp32 run
    is 1e-3..3e-1 away from float64).  Integer argmax coordinates must equal the float64
    oracle's wherever the reference's fp32 run also does."""
    from litehandnet_amd import get_loss, get_model, heatmap
    g = np.load(os.path.join(golden_dir, f"model_{tag}.npz"))
    cfg = litehandnet_cfg(variant, **kw)
    cfg.MODEL["ca_dropout"] = 0.0
    n, size, seed = int(g["n"]), int(g["size"]), int(g["seed"])
    hs = size // 4
    j = synth.synth_joints(n, 21, size, seed + 1)
    tgt = np.stack([onp.msra_generate_target(a, np.ones_like(a), [size, size], [hs, hs])[0] for a in j])
    tw = torch.from_numpy(g["target_weight"])
    # float64 oracle
    ref = torch_ref.get_model(cfg, p_drop=0.0)
    sd = synth.synth_state_dict(ref, seed)
    ref.load_state_dict(sd)
    ref = ref.double().train()
    y64 = ref(synth.synth_images(n, size, seed).double())
    t64, w64 = torch.from_numpy(tgt).double(), tw.double()
    if y64.dim() == 5:         # stacked hourglass [N, S, K, H, W]: every stack supervised by the same target (see loss.py)
        S = y64.shape[1]
        t64, w64 = t64.unsqueeze(1).expand(-1, S, -1, -1, -1), w64.unsqueeze(1).expand(-1, S, -1, -1)
    l64 = cfg.LOSS.loss_weight[0] * torch_ref.distance_loss(y64, t64, w64)
    l64.backward()
    l64 = l64.detach()
    y64n = y64.detach().numpy()
    scale = np.abs(y64n).max()
    ref32_err = np.abs(g["heatmap"] - y64n).max() / scale
    # HIP
    m = get_model(cfg)
    m.load_state_dict(sd)
    m.to(dev).train()
    y = m(synth.synth_images(n, size, seed).to(dev))
    err = np.abs(y.detach().cpu().numpy() - y64n).max() / scale
    assert err <= max(3 * ref32_err, 1e-4), (err, ref32_err)
    loss, _ = get_loss(cfg)(y, {"target": torch.from_numpy(tgt), "target_weight": tw})
    assert abs(float(loss.detach()) - float(l64)) <= max(3 * abs(float(g["loss"]) - float(l64)), 1e-5 * abs(float(l64)))
    loss.backward()
    gn32 = dict(zip(g["grad_keys"].tolist(), g["grad_norms"].tolist()))
    gn64 = {k: float(p.grad.norm()) for k, p in ref.named_parameters()}
    floor = 1e-3 * max(gn64.values())
    errs = {k: abs(float(p.grad.norm()) - gn64[k]) / (gn64[k] + floor) for k, p in m.named_parameters()}
    errs32 = {k: abs(gn32[k] - gn64[k]) / (gn64[k] + floor) for k in gn64}
    # A per-channel SCALAR in front of a train-mode BatchNorm (the 1x1 depthwise branch of RepBlock, repblocks.py:104-107:
    # weight [C, 1, 1, 1]) is almost scale-invariant: BatchNorm undoes the scale, only eps keeps the gradient from being exactly
    # zero.  Its value (0.5 % of the largest gradient norm in model_A_256) is what is left of a 32k-term sum of products that
    # cancel -- any implementation returns it with the absolute error of the terms, not of the result.  Measured for
    # pre.conv1.1.rbr_1x1.conv.weight (lhn_conv_dw_bwd with k = 1: k_dw_bwd_weight<1>, dy formed on the fly as A du + B y + C):
    # 0.5e-4 .. 1.1e-4 of the largest gradient norm depending on the arrival order of the atomics, torch's fp32 run 0.15e-4.
    # Such parameters are held to an ABSOLUTE bar, 3e-4 of the largest gradient norm, instead of a multiple of torch's luck.
    gmax = max(gn64.values())
    zero = [k for k, p in m.named_parameters() if p.dim() == 4 and tuple(p.shape[1:]) == (1, 1, 1) and k.endswith("rbr_1x1.conv.weight")
            and gn64[k] <= 2e-2 * gmax]                                         # nearly cancelled (not with every weight set)
    for k in zero:
        assert errs[k] * (gn64[k] + floor) <= 3e-4 * gmax, (k, errs[k] * (gn64[k] + floor) / gmax)
    worst, worst32 = max(v for k, v in errs.items() if k not in zero), max(v for k, v in errs32.items() if k not in zero)
    top = sorted((k for k in errs if k not in zero), key=lambda k: -errs[k])[:3]
    parity_record(f"model_{tag}_golden", grad_norm_worst=worst, grad_norm_ref_fp32_worst=worst32, grad_norm_bar=max(MODEL_GRAD_FACTOR * worst32, MODEL_GRAD_FLOOR),
                  grad_norm_worst_params=[f"{k}: hip {errs[k]:.3e} / reference-fp32 {errs32[k]:.3e}" for k in top],
                  scale_invariant_params=[f"{k}: abs err hip {errs[k] * (gn64[k] + floor) / gmax:.2e} / reference-fp32 "
                                          f"{errs32[k] * (gn64[k] + floor) / gmax:.2e} of the largest gradient norm (own norm {gn64[k] / gmax:.2e})" for k in zero])
    # whole-network gradient norms are chaotic at this level (one ReLU derivative flipping moves every upstream norm by ~0.4 %,
    # DESIGN.md section 2): any change of summation order -- another kernel for one layer, another atomic arrival order -- moves
    # the worst parameter between 2x and 4x the reference's own fp32 error.  The tight evidence is block-level (_check_block)
    assert worst <= max(MODEL_GRAD_FACTOR * worst32, MODEL_GRAD_FLOOR), (worst, worst32, top)
    bk = str(g["bn_key"])
    rm64 = ref.state_dict()[bk].numpy()
    rm_tol = max(1e-5, 3 * np.abs(g["bn_running_mean"] - rm64).max())
    assert np.abs(m.state_dict()[bk].cpu().numpy() - rm64).max() <= rm_tol
    last = (lambda a: a[:, -1]) if y64n.ndim == 5 else (lambda a: a)       # hourglass: the last stack is the prediction
    p, _ = heatmap._get_max_preds(last(y.detach()).contiguous())
    p64, _ = onp.get_max_preds(np.ascontiguousarray(last(y64n)).astype(np.float32))
    p32, _ = onp.get_max_preds(np.ascontiguousarray(last(g["heatmap"])))
    same32 = (p32 == p64).all(-1)
    assert (p.cpu().numpy() == p64).all(-1)[same32].all()
    pn_ = p.cpu().numpy()
    parity_record(f"model_{tag}_golden", heatmap_err=err, heatmap_err_ref_fp32=ref32_err, heatmap_bar=max(3 * ref32_err, 1e-4),
                  loss_err=abs(float(loss.detach()) - float(l64)) / abs(float(l64)), loss_err_ref_fp32=abs(float(g["loss"]) - float(l64)) / abs(float(l64)),
                  argmax_disagree_vs_f64=int((~(pn_ == p64).all(-1)).sum()), argmax_disagree_vs_ref_fp32=int((~(pn_ == p32).all(-1)).sum()),
                  argmax_ref_fp32_disagree_vs_f64=int((~same32).sum()), keypoints=int(p64.shape[0] * p64.shape[1]))
    print(f"[{tag}] heatmap err vs f64: hip {err:.2e} / reference-fp32 {ref32_err:.2e}; grad-norm err: hip {worst:.2e} / "
          f"reference-fp32 {worst32:.2e}; argmax agree {float((p.cpu().numpy() == p64).all(-1).mean()):.4f}")


def test_model_B_64_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "B_64")


def test_model_B_256_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "B_256")


def test_model_Bca_64_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "Bca_64", rbu_ca="ca")


def test_model_A_64_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "A_64", variant="A")


def test_model_A_256_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "A_256", variant="A")


# ---- variant A blocks (liteHandNet.py)
@pytest.mark.parametrize("c,stride", [(32, 1), (64, 1), (128, 1), (32, 2), (128, 2)])
def test_repconv_dense3x3(dev, c, stride):
    from litehandnet_amd import repblocks
    _check_block(repblocks.RepConv(c, c, 3, stride, 1, activation=torch.nn.LeakyReLU, inplace=True),
                 torch_ref.RepConv(c, c, 3, stride, 1, activation=torch.nn.LeakyReLU, inplace=True), _x(3, c, 12, 16), dev)


def test_repconv_pointwise_stride2(dev):
    from litehandnet_amd import repblocks
    _check_block(repblocks.RepConv(64, 64, 1, 2, 0, activation=None), torch_ref.RepConv(64, 64, 1, 2, 0, activation=None),
                 _x(3, 64, 12, 16), dev)


def test_repblock_dw7_identity(dev):
    from litehandnet_amd import repblocks
    _check_block(repblocks.RepBlock(32, 32, 7, 1, 3, groups=32), torch_ref.RepBlock(32, 32, 7, 1, 3, groups=32),
                 _x(2, 32, 20, 20), dev)


def test_dwconv_A(dev):
    from litehandnet_amd import liteHandNet as la
    _check_block(la.DWConv(64, 32, padding=2, dilation=2), torch_ref.DWConv(64, 32, 2, 2), _x(2, 64, 16, 16), dev)


@pytest.mark.parametrize("red", [2, 4])
def test_bottleneck(dev, red):
    from litehandnet_amd import liteHandNet as la
    _check_block(la.BottleNeck(128, red), torch_ref.BottleNeck(128, red), _x(4, 128, 8, 8), dev)


@pytest.mark.parametrize("stride", [1, 2])
def test_basicblock(dev, stride):
    from litehandnet_amd import liteHandNet as la
    _check_block(la.BasicBlock(128, 128, stride), torch_ref.BasicBlock(128, 128, stride), _x(4, 128, 8, 8), dev)


@pytest.mark.parametrize("ca", ["none", "ca"])
def test_msab(dev, ca):
    from litehandnet_amd import liteHandNet as la
    _check_block(la.MSAB(128, 128, ca, p_drop=0.0), torch_ref.MSAB(128, 128, ca, p_drop=0.0), _x(4, 128, 16, 16), dev)


def test_stem_A(dev):
    from litehandnet_amd import liteHandNet as la
    _check_block(la.Stem(128), torch_ref._StemA(128, torch.nn.LeakyReLU), _x(2, 3, 64, 64), dev, no_dx=True)


def test_state_dict_contract_A(dev):
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg("A")
    ours, ref = get_model(cfg), torch_ref.get_model(cfg)
    assert list(ours.state_dict()) == list(ref.state_dict())
    assert sum(p.numel() for p in ours.parameters()) == 2272981      # test_models_performance.ipynb:247


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


# ---------------------------------------------------------------- deploy-time re-parameterisation (section 8f rank 2)
@pytest.mark.parametrize("variant", ["A", "B"])
def test_deploy_model_golden(dev, golden_dir, variant):
    """deploy_model(): the fused tensors match the reference's (fixture written by make_golden.py from the real reference
    on the CPU) to 2 ulp -- torch's CPU sqrt is not correctly rounded, see oracle.torch_ref.fold_bn_np, against which the
    kernel is bit-exact below --, the parameter count is the notebook's known answer, and the deployed forward matches."""
    from litehandnet_amd import get_model
    g = np.load(os.path.join(golden_dir, f"model_{variant}_64_deploy.npz"))
    m = get_model(litehandnet_cfg(variant))
    m.load_state_dict(synth.synth_state_dict(m, int(g["weights_seed"])))
    m.to(dev).eval()
    x = synth.synth_images(2, 64, int(g["seed"])).to(dev)
    with torch.no_grad():
        y_eval = m(x)
    m.deploy_model()
    sd = m.state_dict()
    assert list(sd) == [str(k) for k in g["keys"]]
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    if variant == "A":
        assert int(g["n_params"]) == 2265621                         # test_models_performance.ipynb:252-254
    for k in g.files:
        if k.startswith("t_"):
            a = sd[k[2:]].cpu().numpy()
            assert np.all(np.abs(a - g[k]) <= 2.4e-7 * np.abs(g[k]) + 3e-7 * np.abs(g[k]).max()), k   # ~2 ulp (stated above)
    sums = np.array([float(v.double().abs().sum()) for v in sd.values()])
    assert np.allclose(sums, g["abs_sums"], rtol=2e-7, atol=0)       # every tensor, via its |.| checksum
    with torch.no_grad():
        y = m(x)                                                     # grad mode on or off: deploy form is inference-only
    ref = g["heatmap"]
    assert np.abs(y.cpu().numpy() - ref).max() <= FWD_TOL * np.abs(ref).max()
    assert _rel(y, y_eval) < 1e-4                                    # same function as the eval-mode train form
    m.deploy_model()                                                 # idempotent
    assert list(m.state_dict()) == [str(k) for k in g["keys"]]


def test_deploy_units_vs_oracle(dev):
    """RepConv / RepBlock (dense, depthwise 7x7, stride 2 without identity) / ChannelAttension one by one."""
    from litehandnet_amd.common import ChannelAttension
    from litehandnet_amd.repblocks import RepBlock, RepConv
    cases = [
        (RepConv(32, 64, 1), torch_ref.RepConv(32, 64, 1), 32),
        (RepConv(32, 32, 3, 1, 2, 2, groups=32, activation=None), torch_ref.RepConv(32, 32, 3, 1, 2, 2, groups=32, activation=None), 32),
        (RepBlock(32, 32, 3, 1, 1), torch_ref.RepBlock(32, 32, 3, 1, 1), 32),
        (RepBlock(32, 32, 7, 1, 3, groups=32), torch_ref.RepBlock(32, 32, 7, 1, 3, groups=32), 32),
        (RepBlock(32, 64, 3, 2, 1), torch_ref.RepBlock(32, 64, 3, 2, 1), 32),
        (ChannelAttension(64, p_drop=0.0), torch_ref.ChannelAttension(64, 0.0), 64),
    ]
    for j, (ours, ref, cin) in enumerate(cases):
        sd = synth.synth_state_dict(ref, 40 + j)
        ref.load_state_dict(sd); ours.load_state_dict(sd)
        ours.to(dev).eval(); ref.eval()
        ours.switch_to_deploy(); ref.switch_to_deploy()
        so, sr = ours.state_dict(), ref.state_dict()
        assert list(so) == list(sr)
        for k in sr:
            assert torch.allclose(so[k].cpu(), sr[k], rtol=2.4e-7, atol=3e-7 * float(sr[k].abs().max())), (j, k)   # sums of 3 branches cancel
        # bit-exact against the IEEE float32 (numpy) evaluation of the reference's formula, branches in its order
        n = lambda t: sd[t].numpy()
        bnp = lambda pre: (n(pre + ".weight"), n(pre + ".bias"), n(pre + ".running_mean"), n(pre + ".running_var"), 1e-5)
        if "rep_conv.weight" in so:
            kw, kb = torch_ref.fold_bn_np(n("conv.conv.weight"), *bnp("conv.bn"))
            name = "rep_conv"
        elif "conv3x3.conv.weight" in sd:
            kw, kb = torch_ref.fold_bn_np(n("conv3x3.conv.weight"), *bnp("conv3x3.bn"))
            name = "rbr_reparam"
        else:
            k3, b3 = torch_ref.fold_bn_np(n("rbr_dense.conv.weight"), *bnp("rbr_dense.bn"))
            k1, b1 = torch_ref.fold_bn_np(n("rbr_1x1.conv.weight"), *bnp("rbr_1x1.bn"))
            pad = k3.shape[-1] // 2
            rest, restb = np.pad(k1, [(0, 0), (0, 0), (pad, pad), (pad, pad)]), b1
            if "rbr_identity.weight" in sd:
                eye = np.zeros_like(k3)
                eye[np.arange(k3.shape[0]), np.arange(k3.shape[0]) % k3.shape[1], pad, pad] = 1
                ki, bi = torch_ref.fold_bn_np(eye, *bnp("rbr_identity"))
                rest, restb = rest + ki, restb + bi
            kw, kb, name = k3 + rest, b3 + restb, "rbr_reparam"
        assert np.array_equal(so[name + ".weight"].cpu().numpy(), kw), (j, "weight vs IEEE fold")
        assert np.array_equal(so[name + ".bias"].cpu().numpy(), kb), (j, "bias vs IEEE fold")
        x = synth.synth_images(2, 16, 50 + j)[:, :1].repeat(1, cin, 1, 1) * torch.linspace(0.5, 1.5, cin).view(1, -1, 1, 1)
        with torch.no_grad():
            yr = ref.double()(x.double())
            yo = ours(x.to(dev))
        assert _rel(yo, yr) < FWD_TOL, (j, _rel(yo, yr))


# ---------------------------------------------------------------- `mynet` (models/pose_hg_ms_att.py, section 8 row a13)
def test_mynet_blocks(dev):
    """Blocks of pose_hg_ms_att.py one by one: biased conv + BN (bias lives in the BN finalize), BN -> SiLU -> conv,
    pooled-BN-ReLU-dw3x3-Linear attention."""
    from litehandnet_amd import pose_hg_ms_att as pm
    _check_block(pm.DWConv(64, 32, padding=2, dilation=2), torch_ref.MyDWConv(64, 32, 2, 2), _x(2, 64, 16, 16), dev)
    _check_block(pm.BottleNeck(128), torch_ref.MyBottleNeck(128), _x(4, 128, 8, 8), dev, seed=1)
    for stride in (1, 2):
        _check_block(pm.BasicBlock(128, 128, stride), torch_ref.MyBasicBlock(128, 128, stride), _x(4, 128, 8, 8), dev, seed=2)
    _check_block(pm.BRC(64, 32, 1, 1, 0), torch_ref.BRC(64, 32, 1, 1, 0), _x(4, 64, 8, 12), dev, seed=3)
    _check_block(pm.ME_att(128, 128, p_drop=0.0), torch_ref.ME_att(128, 128, 0.0), _x(4, 128, 16, 16), dev, seed=4)
    _check_block(pm.my_pelee_stem(128), torch_ref._StemM(128), _x(2, 3, 64, 64), dev, seed=5, no_dx=True)


def test_mynet_contract(dev):
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg("M")
    ours, ref = get_model(cfg), torch_ref.get_model(cfg)
    assert list(ours.state_dict()) == list(ref.state_dict())
    for (k, a), (_, b) in zip(ours.state_dict().items(), ref.state_dict().items()):
        assert a.shape == b.shape, k
    assert sum(p.numel() for p in ours.parameters()) == 2240405       # test_models_performance.ipynb
    # output_acitivation=True (the reference's spelling, pose_hg_ms_att.py:232,251-252): leaky_relu(preds, 0.5) on the head
    cfg.MODEL["output_acitivation"] = True
    cfg.MODEL["ca_dropout"] = 0.0
    ours, ref = get_model(cfg), torch_ref.get_model(cfg, p_drop=0.0)
    _check_block(ours, ref, synth.synth_images(4, 128, 3), dev, seed=52, no_dx=True, grad_tol=2e-2)


def test_model_M_128_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "M_128", variant="M")


def test_model_options_golden(dev, golden_dir):
    """activation='silu' in variant A and mynet's output_acitivation head against the REAL reference's vectors
    (tests/golden/make_golden_extra.py)."""
    _model_case(dev, golden_dir, "Asilu_64", variant="A", activation="silu")
    _model_case(dev, golden_dir, "Mact_128", variant="M", output_acitivation=True)


def test_model_M_256_golden(dev, golden_dir):
    _model_case(dev, golden_dir, "M_256", variant="M")


def test_model_M_eval_golden(dev, golden_dir):
    """Eval mode exercises the conv-bias-inside-running-mean path: shift = beta - (running_mean - bias) * scale."""
    from litehandnet_amd import get_model
    g = np.load(os.path.join(golden_dir, "model_M_64_eval.npz"))
    m = get_model(litehandnet_cfg("M"))
    m.load_state_dict(synth.synth_state_dict(m, int(g["weights_seed"])))
    m.to(dev).eval()
    with torch.no_grad():
        y = m(synth.synth_images(2, 64, int(g["seed"])).to(dev))
    assert np.abs(y.cpu().numpy() - g["heatmap"]).max() <= FWD_TOL * np.abs(g["heatmap"]).max()


def test_hipgraph_replay_matches_plain_launches(dev):
    """LHN_GRAPH=1: lhn_plan_run captures the launch sequence into a hipGraph on its second sighting and replays it.
    Five training iterations on a side stream must give bit-identical forward sums; gradient sums agree to fp32
    atomics noise (weight-gradient replicas are combined by atomics in either mode)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("0", "1"):
        env = dict(os.environ, LHN_GRAPH=mode)
        out = subprocess.run([sys.executable, os.path.join(root, "tests", "check_graph.py")], env=env, capture_output=True,
                             text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("CHECK")][0]
        res[mode] = [tuple(float(v) for v in t.split("/")) for t in line.split()[1:]]
    for (y0, g0), (y1, g1) in zip(res["0"], res["1"]):
        assert y0 == y1
        assert abs(g0 - g1) <= 1e-5 * abs(g0)


@pytest.mark.parametrize("variant", ["A", "B", "M"])
def test_model_224_input(dev, variant):
    """config/litehandnet/freihand/_1_freihand2d_224x224.py trains at 224x224 (56x56 heatmaps, 7x7 lowest level: odd maps,
    ceil-mode pooling, widths below the LDS-tile minimum).  Forward + backward vs the float64 oracle, same bar as the blocks."""
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg(variant, image_size=224)
    cfg.MODEL["ca_dropout"] = 0.0
    ours, ref = get_model(cfg), torch_ref.get_model(cfg, p_drop=0.0)
    # N=4 attention BatchNorms are ill-conditioned (see test_odd_batches): torch's own fp32 run of mynet is 2.7 % away from
    # float64 on the stem's BatchNorm weight there, and which side of the 3x bar a second fp32 implementation lands on
    # depends on its summation order -- mynet runs the case at N=8
    x = synth.synth_images(8 if variant == "M" else 4, 224, 21)
    _check_block(ours, ref, x, dev, seed=30, no_dx=True, grad_tol=2e-2)


# ---------------------------------------------------------------- SyncBatchNorm (section 8f rank 3)
@pytest.mark.parametrize("variant", ["B", "M"])
def test_sync_batchnorm_two_emulated_ranks(dev, variant):
    """cfg.TRAIN.syncBN (train/spawn_dist.py:37-38): two replicas, each with half of a batch of 8, run in lockstep on one
    GPU (two host threads; the all-reduce of every statistics buffer is emulated by summing the two replicas' buffers).
    SyncBatchNorm over the halves must equal plain BatchNorm over the WHOLE batch as computed by the float64 ORACLE (not
    by this library): outputs, running statistics, and the SUM of the two ranks' parameter gradients (DDP would average
    them).  Yardstick: the oracle's own fp32 CPU run of the whole batch."""
    import threading
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg(variant)
    cfg.MODEL["ca_dropout"] = 0.0
    ref = torch_ref.get_model(cfg, p_drop=0.0)
    sd = synth.synth_state_dict(ref, 17)
    ref.load_state_dict(sd)
    ref.train()
    x = synth.synth_images(8, 64, 5)
    g = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).standard_normal((8, 21, 16, 16)).astype(np.float32))
    ref32 = copy.deepcopy(ref)
    y32 = ref32(x)
    y32.backward(g)
    ref = ref.double()
    yf = ref(x.double())
    yf.backward(g.double())
    gf = {k: p.grad.clone() for k, p in ref.named_parameters()}
    g32 = {k: p.grad.clone() for k, p in ref32.named_parameters()}
    full = ref
    x, g = x.to(dev), g.to(dev)

    world = 2
    reps = []
    for r in range(world):
        m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(get_model(cfg))
        m.load_state_dict(sd)
        m.to(dev).train()
        reps.append(m)
    barrier, slots, errors, outs = threading.Barrier(world), [None] * world, [], [None] * world

    def make_allreduce(rank):
        def allreduce(t):
            slots[rank] = t
            barrier.wait()
            total = slots[0] + slots[1]          # same stream: ordered after both ranks' producer kernels
            barrier.wait()
            t.copy_(total)
            barrier.wait()
        return allreduce

    def worker(rank):
        try:
            with torch.autograd.set_multithreading_enabled(False):
                m = reps[rank]
                from litehandnet_amd.engine import Engine
                eng = Engine(m)
                m.__dict__["_engine"] = eng
                eng.sync_override = (world, make_allreduce(rank))
                y = m(x[4 * rank:4 * rank + 4])
                y.backward(g[4 * rank:4 * rank + 4])
                outs[rank] = y.detach()
        except Exception as e:          # noqa: BLE001 -- surfaced below
            errors.append(repr(e))
            barrier.abort()

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=240)
    assert not errors, errors
    y2 = torch.cat(outs)
    assert _rel(y2, yf) < max(FWD_TOL, 3 * _rel(y32, yf)), (_rel(y2, yf), _rel(y32, yf))
    pa, pb2 = dict(reps[0].named_parameters()), dict(reps[1].named_parameters())
    floor = 1e-3 * max(float(v.double().norm()) for v in gf.values())
    e32s = {k: float((g32[k].double() - v).norm() / (v.norm() + floor)) for k, v in gf.items()}
    for k, v in gf.items():
        tot = pa[k].grad.cpu().double() + pb2[k].grad.cpu().double()
        e = float((tot - v).norm() / (v.norm() + floor))
        assert e < max(GRAD_TOL, 3 * e32s[k], 1.5 * max(e32s.values())), (k, e, e32s[k])
    for (k, a), (_, b) in zip(reps[0].state_dict().items(), full.state_dict().items()):
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.allclose(a.cpu().double(), b, rtol=1e-4, atol=1e-5), k       # the bar of _check_block


# ---------------------------------------------------------------- squeeze-and-excitation gates ('se')
def test_se_blocks_and_model(dev, golden_dir):
    """ca_type / msrb_ca / rbu_ca = 'se' (common.py:23-37; config/litehandnet/*_h4_se_none.py): the blocks against the
    float64 oracle, the whole variant-B model against the reference-generated fixture."""
    from litehandnet_amd import litehourglass as lh, liteHandNet as la
    from litehandnet_amd.common import ChannelAttension, SEBlock
    _check_block(SEBlock(64, 4), torch_ref.SEBlock(64, 4), _x(4, 64, 12, 12), dev, seed=60)
    _check_block(ChannelAttension(64, p_drop=0.0), torch_ref.ChannelAttension(64, 0.0), _x(8, 64, 12, 12), dev, seed=64)
    _check_block(lh.MSRB(64, 64, "se", p_drop=0.0), torch_ref.MSRB(64, 64, "se", 0.0), _x(4, 64, 16, 16), dev, seed=61)
    _check_block(lh.RepBasicUnit(64, 64, "se", p_drop=0.0), torch_ref.RepBasicUnit(64, 64, "se", 0.0), _x(4, 64, 12, 12), dev, seed=62)
    _check_block(la.MSAB(128, 128, "se", p_drop=0.0), torch_ref.MSAB(128, 128, "se", p_drop=0.0), _x(4, 128, 16, 16), dev, seed=63)
    _model_case(dev, golden_dir, "Bse_64", variant="B", msrb_ca="se", rbu_ca="se")


@pytest.mark.parametrize("variant,n,size", [("B", 3, 128), ("A", 5, 96), ("M", 1, 96), ("B", 1, 256)])
def test_odd_batches(dev, variant, n, size):
    """Batch sizes / map sizes that leave partially filled GEMM tiles everywhere (N*H*W not a multiple of 64 or 128 on the
    low-resolution levels; N = 1).  Forward + backward against the float64 oracle.  (N = 1 in training mode makes the
    attention's BatchNorm over N samples degenerate -- eval-mode forward only there.)

    Gradient bar 3e-2 instead of the blocks' 1e-3: with ReLU-like units a single pre-activation within one fp32 ulp of
    zero flips its derivative relative to float64 (traced on the A / N=4 / 96px case: ONE (pixel, channel) of one layer,
    its 3x3 dgrad footprint 62 % off, every upstream gradient norm moved by ~0.4 %); which element flips depends on the
    summation order, so torch's own fp32 run may or may not share it."""
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg(variant, image_size=size)
    cfg.MODEL["ca_dropout"] = 0.0
    ours, ref = get_model(cfg), torch_ref.get_model(cfg, p_drop=0.0)
    x = synth.synth_images(n, size, 40 + n)
    if n > 1:
        _check_block(ours, ref, x, dev, seed=70 + n, no_dx=True, grad_tol=3e-2)
        return
    sd = synth.synth_state_dict(ref, 71)
    ref.load_state_dict(sd); ours.load_state_dict(sd)
    ours.to(dev).eval(); ref.double().eval()
    with torch.no_grad():
        y, y64 = ours(x.to(dev)), ref(x.double())
    assert _rel(y, y64) < FWD_TOL, _rel(y, y64)


def test_width_256(dev):
    """input_channel = 256 -- the reference's own default (litehourglass.py:202) and config/mynet/_7_*_c256.py: 1x1 GEMMs
    with K and N = 256 run as 128-wide slices inside the library.  Variant B forward + backward vs the float64 oracle."""
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg("B", channels=256)
    cfg.MODEL["ca_dropout"] = 0.0
    ours, ref = get_model(cfg), torch_ref.get_model(cfg, p_drop=0.0)
    _check_block(ours, ref, synth.synth_images(4, 64, 23), dev, seed=24, no_dx=True, grad_tol=2e-2)


def test_unsupported_width_fails_loudly(dev):
    """A width whose halves are not multiples of 4 channels (input_channel = 100: RepBasicUnit splits 50 / 50) must raise
    LhnError on the first forward -- never fall back to anything else.  (Widths like 96 run: any multiple of 8.)"""
    from litehandnet_amd import _lib, get_model
    m = get_model(litehandnet_cfg("B", channels=100)).to(dev).train()
    with pytest.raises(_lib.LhnError, match="multiples of 4"):
        m(torch.zeros(2, 3, 64, 64, device=dev))
    m96 = get_model(litehandnet_cfg("B", channels=96)).to(dev).train()
    with torch.no_grad():
        assert torch.isfinite(m96(torch.randn(2, 3, 64, 64, device=dev))).all()


def test_unsupported_inputs_fail_loudly(dev):
    """No silent wrong answers at the module boundary: image gradients, reduced-precision parameters, CPU tensors."""
    from litehandnet_amd import _lib, get_model
    cfg = litehandnet_cfg("B")
    m = get_model(cfg).to(dev).train()
    with pytest.raises(_lib.LhnError, match="input image"):
        m(torch.zeros(2, 3, 64, 64, device=dev, requires_grad=True))
    with pytest.raises(_lib.LhnError, match="float32"):
        m(torch.zeros(2, 3, 64, 64, device=dev, dtype=torch.float16))
    with pytest.raises(_lib.LhnError):
        m(torch.zeros(2, 3, 64, 64))                      # CPU input: there is no CPU path
    h = get_model(cfg).to(dev).half()
    with pytest.raises(_lib.LhnError, match="float32"):
        h(torch.zeros(2, 3, 64, 64, device=dev))


def test_eval_tables_are_cached_and_invalidated(dev):
    """Eval / deployed forwards skip the BatchNorm-table launches when nothing changed (LHN_RUN_TABLES_CURRENT); any change of
    a parameter or running statistic -- through torch or through our own train-mode kernels -- must rebuild them."""
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg("B")
    cfg.MODEL["ca_dropout"] = 0.0
    m = get_model(cfg)
    m.load_state_dict(synth.synth_state_dict(torch_ref.get_model(cfg, p_drop=0.0), 5))
    m.to(dev).eval()
    x = synth.synth_images(2, 64, 11).to(dev)

    def fresh():
        m.__dict__.pop("_engine", None)            # new engine = new plan = tables rebuilt from scratch
        with torch.no_grad():
            return m(x).clone()
    with torch.no_grad():
        y1 = m(x).clone()
        y2 = m(x).clone()                          # second run: tables reused
    assert torch.equal(y1, y2)
    plan = next(iter(m.__dict__["_engine"].plans.values()))
    assert plan._table_sig is not None
    # 1. torch-side change of a running statistic
    with torch.no_grad():
        next(b for k, b in m.named_buffers() if k.endswith("running_mean")).add_(0.25)
        y3 = m(x).clone()
    assert not torch.equal(y3, y1)
    assert torch.equal(y3, fresh())
    # 2. our own train-mode kernels move the running statistics behind torch's back
    with torch.no_grad():
        y4 = m(x).clone()
        y4b = m(x).clone()
    assert torch.equal(y4, y4b)
    m.train()
    with torch.no_grad():
        m(synth.synth_images(4, 64, 12).to(dev))
    m.eval()
    with torch.no_grad():
        y5 = m(x).clone()
    assert not torch.equal(y5, y4)
    assert torch.equal(y5, fresh())
    # 3. deployed form: biases live in the same tables
    m.deploy_model()
    with torch.no_grad():
        y6 = m(x).clone()
        y7 = m(x).clone()
    assert torch.equal(y6, y7)
    assert float((y6 - y5).abs().max()) <= 1e-3 * float(y5.abs().max())


@pytest.mark.parametrize("kind", ["pw", "dw1", "dw3", "dense3x3"])
def test_batchnorm_statistics_large_mean(dev, kind):
    """BatchNorm batch statistics when |mean| = 1000 x sigma at the convolution output (E[x^2] - E[x]^2 cancels 6 digits):
    the epilogues promote their per-tile fp32 partial sums to double.  Checked where it shows: the batch VARIANCE (through
    running_var, momentum 0.1 on an initial value 1) within 1e-3 of the float64 oracle -- a running fp32 sum of squares over
    the 16K pixels per channel of this test is off by tens of percent -- and the normalised output no further from float64
    than 3x the oracle's own fp32 CPU run (the convolution output itself only resolves 1000 +- 1 to ~1e-4)."""
    from litehandnet_amd import repblocks
    r = np.random.Generator(np.random.PCG64(77))
    c = 64
    mk = {"pw": lambda M: M.RepConv(c, c, 1, activation=None), "dw1": lambda M: M.RepConv(c, c, 1, groups=c, activation=None),
          "dw3": lambda M: M.RepConv(c, c, 3, 1, 1, groups=c, activation=None), "dense3x3": lambda M: M.RepConv(c, c, 3, 1, 1, activation=None)}[kind]
    ours, ref = mk(repblocks), mk(torch_ref)
    sd = synth.synth_state_dict(ref, 3)
    w = sd["conv.conv.weight"]
    if kind == "pw":
        w = w - w.mean(dim=(1, 2, 3), keepdim=True) + 1.0 / w[0].numel()          # every output feature: weights sum to 1
    else:                                                                          # (near-)identity kernels: no border effect
        d = torch.zeros_like(w)
        k = w.shape[-1] // 2
        if w.shape[1] == 1:
            d[:, 0, k, k] = 1.0
        else:
            d[torch.arange(c), torch.arange(c), k, k] = 1.0
        w = d + 1e-5 * w
    sd["conv.conv.weight"] = w
    sd["conv.bn.running_var"] = torch.ones_like(sd["conv.bn.running_var"])
    ref.load_state_dict(sd); ours.load_state_dict(sd)
    x = torch.from_numpy((1000.0 + r.standard_normal((4, c, 64, 64))).astype(np.float32))
    ref32 = copy.deepcopy(ref).train()
    ours.to(dev).train(); ref.double().train()
    with torch.no_grad():
        y, y64, y32 = ours(x.to(dev)), ref(x.double()), ref32(x)
    pad = 0 if w.shape[-1] == 1 else 1
    pre = torch.nn.functional.conv2d(x.double(), w.double(), padding=pad, groups=(c if w.shape[1] == 1 else 1))
    ratio = float((pre.mean(dim=(0, 2, 3)).abs() / pre.std(dim=(0, 2, 3))).min())
    assert ratio > 300, ratio                                    # the premise: mean >> sigma at the BatchNorm input
    rv = dict(ours.named_buffers())["conv.bn.running_var"].cpu().double()
    rv64 = dict(ref.named_buffers())["conv.bn.running_var"]
    bvar, bvar64 = (rv - 0.9) / 0.1, (rv64 - 0.9) / 0.1            # the batch variance behind the momentum update
    assert float(((bvar - bvar64).abs() / bvar64).max()) < 1e-3, (kind, float(((bvar - bvar64).abs() / bvar64).max()))
    e, e32 = _rel(y, y64), _rel(y32, y64)
    assert e < max(5e-4, 3 * e32), (kind, e, e32, ratio)



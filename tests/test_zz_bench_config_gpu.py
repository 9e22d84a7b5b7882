"""BASELINE configs 2 and 3 at FULL size, exactly what bench.py times (batch 64, 256x256, train-mode BatchNorm, Dropout2d 0.3).
Sorted last on purpose: the CPU oracle runs of this file (float64 arbiter + fp32 yardstick, 1-2 minutes per variant) are
computed by a child process that conftest.py starts at the beginning of the GPU session (tests/bench_config_oracle.py), so
they overlap the rest of the suite instead of adding to it."""
import numpy as np
import pytest
import torch

from conftest import bench_oracle, parity_record
from litehandnet_amd.config import litehandnet_cfg
from oracle import heatmap_np as onp
from oracle import synth, torch_ref
from test_dropout_gpu import P, _attach

pytestmark = pytest.mark.gpu


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
    ours = get_model(cfg)
    ours.load_state_dict(synth.synth_state_dict(torch_ref.get_model(cfg, p_drop=P), seed))
    ours.to(dev).train()
    masks = _attach(ours, n, seed + 500)
    x = synth.synth_images(n, size, seed)
    j = synth.synth_joints(n, 21, size, seed + 1)
    tgt = torch.from_numpy(np.stack([onp.msra_generate_target(a, np.ones_like(a), [size, size], [64, 64])[0] for a in j]))
    tw = torch.ones(n, 21, 1)
    y = ours(x.to(dev))
    loss, _ = get_loss(cfg)(y, {"target": tgt, "target_weight": tw})
    loss.backward()
    # the CPU oracle's side, computed by the child process with the SAME seeds and masks (checked)
    o = bench_oracle(variant)
    assert list(o["mask_names"]) == list(masks) and all(np.array_equal(o[f"mask_{i}"], v.numpy()) for i, v in enumerate(masks.values()))
    y64n, y32, l64, l32 = o["y64"], o["y32"], float(o["l64"]), float(o["l32"])
    g64, g32 = dict(zip(o["keys"].tolist(), o["g64"].tolist())), dict(zip(o["keys"].tolist(), o["g32"].tolist()))
    scale = np.abs(y64n).max()
    e32 = np.abs(y32 - y64n).max() / scale
    err = np.abs(y.detach().cpu().numpy() - y64n).max() / scale
    assert err <= max(1e-4, 3 * e32), (err, e32)
    assert abs(float(loss.detach()) - float(l64)) <= max(3 * abs(float(l32) - float(l64)), 1e-5 * abs(float(l64)))
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
    parity_record(f"bench_config_bs64_256_{variant}", heatmap_err=err, heatmap_err_cpu_fp32=e32, grad_norm_worst=worst, grad_norm_cpu_fp32_worst=worst32,
                  grad_norm_bar=max(MODEL_GRAD_FACTOR * worst32, MODEL_GRAD_FLOOR), argmax_disagree_vs_f64=int(diff.sum()),
                  argmax_disagree_vs_cpu_fp32=int((~(pn == p32).all(-1)).sum()), argmax_cpu_fp32_disagree_vs_f64=int((~same32).sum()),
                  keypoints=int(n * 21), pck_vs_f64_decode=float(avg), pck_delta=float(1.0 - avg))
    print(f"[{variant} bs64 256 p=0.3] heatmap err vs f64: hip {err:.2e} / cpu-fp32 {e32:.2e}; grad-norm: hip {worst:.2e} / "
          f"cpu-fp32 {worst32:.2e}; argmax agree {float((pn == p64).all(-1).mean()):.4f} (fp32 cpu {float(same32.mean()):.4f}), near-ties {int(diff.sum())}; PCK {avg:.4f}")

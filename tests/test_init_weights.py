"""CPU: `init_weights` / `normal_init` semantics pinned against the REAL reference.

tests/golden/init_weights.json holds, per model, the sha256 over every tensor of the reference's `state_dict()` built under
torch.manual_seed(seed) (written by tests/golden/make_golden_r3.py, which imports /root/reference in the build container).
The mirrors (litehandnet_amd.get_model) and the oracle (oracle.torch_ref.get_model) built under the same seed must give the
same bytes: same construction order (same draws from torch's generator), Conv2d weight ~ N(0, 1), bias 0, BatchNorm weight
~ N(0, 1) for the registered `litehandnet` (liteHandNet.py:236-238 + weight_init.py:28-32 touch EVERY module with a
`.weight`), 1 for the MSRB hourglass / mynet / hourglass / Lite-HRNet.  bench.py's "random-init weights" rest on this."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from litehandnet_amd import get_model
from litehandnet_amd.config import litehandnet_cfg
from oracle import torch_ref

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "init_weights.json")))


def _digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("tag", sorted(GOLD["models"]))
def test_initial_state_equals_the_reference(tag):
    e = GOLD["models"][tag]
    cfg = litehandnet_cfg(e["variant"], **e["kw"])
    for build in (get_model, torch_ref.get_model):
        torch.manual_seed(GOLD["seed"])
        m = build(cfg)
        sd = m.state_dict()
        assert len(sd) == e["tensors"] and sum(p.numel() for p in m.parameters()) == e["params"]
        for k, want in e["sums"].items():            # a readable first difference before the digest
            assert float(sd[k].double().sum()) == want, (tag, build.__module__, k)
        assert _digest(sd) == e["sha256"], (tag, build.__module__)


def test_registered_litehandnet_draws_batchnorm_gammas():
    """Variant A: BatchNorm gamma ~ N(0, 1) (about half of them negative), beta 0, running statistics untouched; variant B:
    gamma 1 (litehourglass.py:224-230)."""
    torch.manual_seed(GOLD["seed"])
    a = get_model(litehandnet_cfg("A"))
    gam = torch.cat([m.weight.detach().flatten() for m in a.modules() if isinstance(m, torch.nn.BatchNorm2d)])
    assert abs(float((gam < 0).float().mean()) - 0.5) < 0.05 and abs(float(gam.std()) - 1.0) < 0.05
    assert abs(float((gam < 0).float().mean()) - GOLD["models"]["A"]["bn_gamma_negative_fraction"]) < 0.02
    for m in a.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            assert float(m.bias.abs().max()) == 0 and float(m.running_mean.abs().max()) == 0 and float((m.running_var - 1).abs().max()) == 0
    b = get_model(litehandnet_cfg("B"))
    assert all(float((m.weight - 1).abs().max()) == 0 for m in b.modules() if isinstance(m, torch.nn.BatchNorm2d))
    conv = torch.cat([m.weight.detach().flatten() for m in b.modules() if isinstance(m, torch.nn.Conv2d)])
    assert abs(float(conv.std()) - 1.0) < 0.02 and abs(float(conv.mean())) < 0.02


def test_random_scale_rotation_fixture_is_reference_output():
    """tests/golden/scale_rotation.npz (the reference's TopDownGetRandomScaleRotation under np.random.seed) against the host
    sampler of the GPU input path (litehandnet_amd.pipeline.random_scale_rotation): same numpy draws in the same order."""
    from litehandnet_amd import pipeline
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scale_rotation.npz"))
    np.random.seed(int(g["seed"]))
    s, r = pipeline.random_scale_rotation(g["scale_in"], float(g["rot_factor"]), float(g["scale_factor"]), float(g["rot_prob"]))
    assert np.array_equal(s.astype(np.float64), g["scale_out"]) and np.array_equal(r.astype(np.float64), g["rotation"])

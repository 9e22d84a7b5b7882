"""CPU (no GPU visible): the product path has NO CPU fallback -- every entry point must fail loudly instead of computing
something on the host, and the product package never imports the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from litehandnet_amd import _lib, get_loss, get_model, heatmap
from litehandnet_amd.config import litehandnet_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
no_gpu = pytest.mark.skipif(torch.cuda.is_available(), reason="checks the behaviour of a GPU-less host")


@no_gpu
def test_model_and_trainer_refuse_to_run_without_a_gpu():
    from litehandnet_amd.train import Trainer
    cfg = litehandnet_cfg("B")
    m = get_model(cfg)                                   # building the module tree is host work and is allowed
    with pytest.raises(_lib.LhnError, match="no GPU|GPU"):
        m(torch.zeros(1, 3, 256, 256))
    with pytest.raises(_lib.LhnError, match="no GPU|GPU"):
        Trainer(m, get_loss(cfg))


@no_gpu
def test_heatmap_and_loss_ops_refuse_to_run_without_a_gpu():
    cfg = litehandnet_cfg("B")
    j = np.zeros((2, 21, 3), np.float32)
    hm = np.zeros((2, 21, 64, 64), np.float32)
    c, s = np.zeros((2, 2), np.float32), np.ones((2, 2), np.float32)
    calls = [
        lambda: heatmap.generate_target_batch(j, np.ones_like(j), [256, 256], [64, 64]),
        lambda: heatmap._get_max_preds(hm),
        lambda: heatmap.keypoints_from_heatmaps(hm, c, s),
        lambda: heatmap.keypoints_from_heatmaps(hm, c, s, post_process="unbiased"),
        lambda: heatmap.heatmap_nms(hm),
        lambda: heatmap.keypoint_pck_accuracy(j[..., :2], j[..., :2], np.ones((2, 21), bool), 0.2, np.ones((2, 2), np.float32)),
        lambda: heatmap.generate_simdr_batch(j, np.ones_like(j), [256, 256]),
        lambda: get_loss(cfg)(torch.zeros(2, 21, 64, 64), {"target": torch.zeros(2, 21, 64, 64), "target_weight": torch.ones(2, 21, 1)}),
    ]
    for f in calls:
        with pytest.raises(_lib.LhnError):
            f()


def test_product_package_never_imports_the_oracle():
    """`oracle/` is test infrastructure: importing and building the whole product package must not pull it in."""
    code = ("import sys; sys.path.insert(0, %r); import litehandnet_amd, litehandnet_amd.train, litehandnet_amd.pipeline, "
            "litehandnet_amd.heatmap, litehandnet_amd.loss, litehandnet_amd.models; "
            "from litehandnet_amd.config import litehandnet_cfg; "
            "[litehandnet_amd.get_model(litehandnet_cfg(v)) for v in 'ABM']; "
            "bad = [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]; print('ORACLE', bad)") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    assert "ORACLE []" in out.stdout, out.stdout
    for root, _, files in os.walk(os.path.join(ROOT, "litehandnet_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text, f

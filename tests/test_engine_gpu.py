"""Engine hazards that must fail loudly (one workspace per input shape) and host-side re-entrancy of the library."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

from litehandnet_amd.config import litehandnet_cfg
from oracle import synth

pytestmark = pytest.mark.gpu


def _model(dev):
    from litehandnet_amd import get_model
    cfg = litehandnet_cfg("B")
    cfg.MODEL["ca_dropout"] = 0.0
    return get_model(cfg).to(dev).train()


def test_stale_forward_and_double_backward_raise(dev):
    """Stock torch handles `f(m(x1)) + f(m(x2))`, two forwards then two backwards, or backward(retain_graph=True) twice;
    this engine keeps ONE workspace per input shape, so each of those must raise LhnError instead of silently using the
    wrong activations.  A forward under no_grad (its own inference plan) in between is fine."""
    from litehandnet_amd import _lib
    m = _model(dev)
    x1, x2 = synth.synth_images(2, 64, 1).to(dev), synth.synth_images(2, 64, 2).to(dev)
    y1 = m(x1)
    y2 = m(x2)
    with pytest.raises(_lib.LhnError, match="stale forward"):
        (y1.sum() + y2.sum()).backward()
    m.zero_grad(set_to_none=True)                # (whichever of the two backwards ran first has published its gradients)
    y = m(x1)
    with torch.no_grad():
        m(x2)                                    # inference plan: different workspace
    y.sum().backward(retain_graph=True)
    g1 = next(m.parameters()).grad.clone()
    with pytest.raises(_lib.LhnError, match="second backward"):
        y.sum().backward()
    m.zero_grad(set_to_none=True)
    y = m(x1)
    y.sum().backward()                           # the plain sequence still works and reproduces the gradient
    g2 = next(m.parameters()).grad
    assert float((g2 - g1).norm() / g1.norm()) < 1e-3      # (random-init weights: huge gradients, fp32 atomics order)
    m.eval()
    y = m(x1)                                    # grad-enabled eval forward shares the training plan's workspace ...
    m.train()
    y3 = m(x1)
    m.eval()
    m(x1)
    with pytest.raises(_lib.LhnError, match="stale forward"):
        y3.sum().backward()


def test_two_host_threads_share_the_library(dev):
    """Two host threads (nn.DataParallel style, test.py:81) call C-ABI entry points concurrently on their own streams:
    per-device kernel setup is guarded by call_once, the error channel is thread-local, results equal the serial ones."""
    from litehandnet_amd import _lib, heatmap
    r = np.random.Generator(np.random.PCG64(3))
    hms = [torch.from_numpy(r.random((16, 21, 64, 64)).astype(np.float32)).to(dev) for _ in range(2)]
    want = [heatmap._get_max_preds(h)[0].cpu() for h in hms]
    m = _model(dev).eval()
    x = synth.synth_images(4, 64, 5).to(dev)
    with torch.no_grad():
        yw = m(x).cpu()
    outs, errs = [None, None], []

    def work(i):
        try:
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                for _ in range(20):
                    p, _ = heatmap._get_max_preds(hms[i])
                if i == 0:
                    with torch.no_grad():
                        outs[0] = (p.cpu(), m(x).cpu())
                else:
                    # a failing call on this thread must not leak its message into the other thread's error slot
                    rc = _lib.lib().lhn_heatmap_argmax(None, None, None, None, 0, 0, 0, 0, None)
                    assert rc != 0 and b"lhn_heatmap_argmax" in _lib.lib().lhn_last_error()
                    outs[1] = (p.cpu(), None)
            s.synchronize()
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    assert not errs, errs
    assert torch.equal(outs[0][0], want[0]) and torch.equal(outs[1][0], want[1])
    assert torch.allclose(outs[0][1], yw, rtol=1e-5, atol=1e-6)

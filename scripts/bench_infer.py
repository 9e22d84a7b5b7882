"""Inference latency: eval-mode train form vs deployed (re-parameterised) form.  python scripts/bench_infer.py [A|B] [bs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litehandnet_amd import get_model
from litehandnet_amd.config import litehandnet_cfg

variant = sys.argv[1] if len(sys.argv) > 1 else "B"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = get_model(litehandnet_cfg(variant)).cuda().eval()
x = torch.randn(bs, 3, 256, 256, device="cuda")


side = torch.cuda.Stream()


def timeit(tag):
    with torch.no_grad(), torch.cuda.stream(side):
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(20):
            m(x)
        e1.record()
        torch.cuda.synchronize()
        print(f"{variant} bs{bs} {tag}: {e0.elapsed_time(e1) / 20:.3f} ms/fwd (wall {(time.perf_counter() - t0) * 50:.3f}) "
              f"-> {bs / (e0.elapsed_time(e1) / 20) * 1e3:.0f} img/s", flush=True)


timeit("eval (BN running stats)")
m.deploy_model()
timeit("deployed")

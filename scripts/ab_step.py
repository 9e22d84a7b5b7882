"""A/B timing helper: `python scripts/ab_step.py TAG [variant]` prints TAG, ms per training step and ms per forward of
bench.py's workload (bs64 256x256); environment switches (LHN_*) are read by the library as usual."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench
tag = sys.argv[1]
variant = sys.argv[2] if len(sys.argv) > 2 else "B"
a = argparse.Namespace(gpus=1, steps=30, warmup=8, variant=variant, batch=64, no_cpu_baseline=True, no_extras=True, sync_bn=False,
                       dropout=0.3)
el, fwd = bench.measure_variant(variant, a, torch.device("cuda:0"), 1, 0)
print(tag, "step_ms %.3f fwd_ms %.4f" % (el / a.steps * 1e3, fwd), flush=True)

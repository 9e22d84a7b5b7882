#!/usr/bin/env python3
"""Headline benchmark: images/sec, forward + TopdownHeatmapLoss + backward + Adam step, 256x256, per-GPU batch 64,
litehandnet (MSRB hourglass = variant B by default), synthetic inputs resident in HBM, random-init weights.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...            (no WORLD_SIZE in the environment: starts its own N ranks, see spawn_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (HBM, measured live with
events on the launch stream) and `cpu_baseline` (the CPU oracle = port of the reference's path, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per image, fp32: 4 B x sum over convs of (in + out) elements
ALG_FWD_BYTES = {"B": 76.42e6, "A": 98.07e6, "M": 82.59e6,   # M = `mynet` (pose_hg_ms_att.py), scripts/dump_plan.py M
                 "H": 220.14e6,                              # hourglass, 2 stacks, C = 256 (config/hourglass/_2_*_h2.py)
                 "L": 96.23e6}                               # Lite-HRNet-18 (config/litehrnet/_2_*_18.py)
ALG_FWD_FLOPS = {"B": 0.757e9, "A": 2.560e9, "M": 2.220e9, "H": 16.745e9, "L": 0.634e9}   # conv FLOPs per image, forward
LOSS_BYTES = (8 + 4) * 21 * 64 * 64          # loss fwd reads o,t; bwd writes g (per image)
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable copy rate)
MFMA_FP32_PEAK_TF = 157.3                    # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
VARIANT_NAME = {"B": "MSRB hourglass litehourglass.py", "A": "registered liteHandNet.py, reduction 4", "M": "mynet pose_hg_ms_att.py",
                "H": "stacked hourglass hourglassnet.py, 2 stacks, C=256", "L": "Lite-HRNet-18 lite_hrnet.py"}


def source_sha16():
    """Fingerprint of what decides the forward's HBM traffic (kernels + plan + module mirrors): the PMC figures in
    profiles/pmc_traffic.json carry the fingerprint they were measured at; a different one means they are stale."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pk = os.path.join(ROOT, "litehandnet_amd")
    for f in sorted(glob.glob(os.path.join(pk, "csrc", "*")) + glob.glob(os.path.join(pk, "*.py"))):
        if os.path.isfile(f) and not f.endswith((".so", ".o")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def copy_bandwidth(dev):
    """Measured device-to-device copy rate of THIS GPU (read + write bytes / s) over 1 GiB buffers (the
    empirical roofline denominator next to the 8 TB/s specification)."""
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    gbs = 2 * a.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    return gbs


def cpu_baseline(variant, budget_s=20.0):
    """The reference's path on the host cores: the CPU oracle (fp32 torch, NCHW), fwd + loss + bwd, batch 8."""
    from litehandnet_amd.config import litehandnet_cfg
    from oracle import heatmap_np as onp
    from oracle import synth, torch_ref
    import numpy as np
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))             # a one-GPU box owns a 16-core share of the host
    torch.set_num_threads(cores)
    cfg = litehandnet_cfg(variant)
    m = torch_ref.get_model(cfg)
    m.train()
    crit = torch_ref.TopdownHeatmapLoss(cfg)
    bs = 8
    x = synth.synth_images(bs, 256, 0)
    j = synth.synth_joints(bs, 21, 256, 1)
    t = torch.from_numpy(np.stack([onp.msra_generate_target(a, np.ones_like(a), [256, 256], [64, 64])[0] for a in j]))
    meta = {"target": t, "target_weight": torch.ones(bs, 21, 1)}
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)

    def step():
        y = m(x)
        loss, _ = crit(y, meta)
        opt.zero_grad()
        loss.backward()
        opt.step()
    step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 40 or (n >= 2 and el > 0.5 * budget_s):
            break
    return {"value": round(bs * n / el, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of batch {bs} at 256x256 (fwd+loss+bwd+Adam), torch CPU fp32 oracle, {el:.1f}s"}


def kernel_rooflines(B, dev):
    """Per-kernel HBM rooflines of the two convolution kernels that carry most of the MSRB hourglass' algorithmic
    bytes, at their 64x64 shapes, timed live with events around direct C-ABI launches (stream = torch's current)."""
    import ctypes as C
    from litehandnet_amd import _lib
    from litehandnet_amd._lib import View
    L, st = _lib.lib(), _lib.stream()

    def view(t, table=None):
        v = View()
        v.data, v.table, v.gate = t.data_ptr(), (table.data_ptr() if table is not None else None), None
        v.N, v.H, v.W, v.cstride, v.coff, v.C = t.shape[0], t.shape[1], t.shape[2], t.shape[3], 0, t.shape[3]
        return v

    def timed(fn, reps=30):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    # cache-cold: every launch works on a different (input, output) pair out of NPAIR, 2 x 67 MB each -- 1.07 GB between two
    # uses of the same buffer, far beyond the 256 MiB Infinity Cache (MI355X_MICROARCH.md), so the rate is an HBM rate
    NPAIR = 8
    xs = [torch.randn(B, 64, 64, 64, device=dev) for _ in range(NPAIR)]
    ys = [torch.empty_like(xs[0]) for _ in range(NPAIR)]
    tb = torch.ones(3, 64, device=dev)
    stats = torch.zeros(32 * 2 * 64, dtype=torch.float64, device=dev)
    w3 = torch.randn(64, 1, 3, 3, device=dev)
    w1 = torch.randn(64, 64, device=dev)
    vxs, vys = [view(x, tb) for x in xs], [view(y) for y in ys]
    nbytes = 2 * xs[0].numel() * 4
    turn = [0]

    def dw():
        i = turn[0] = (turn[0] + 1) % NPAIR
        L.lhn_conv_dw_fwd(C.byref(vxs[i]), _lib.ptr(w3), C.byref(vys[i]), _lib.ptr(stats), 3, 1, 1, 1, None, st)

    def pw():
        i = turn[0] = (turn[0] + 1) % NPAIR
        L.lhn_conv_pw_fwd(C.byref(vxs[i]), _lib.ptr(w1), None, C.byref(vys[i]), _lib.ptr(stats), 1, None, None, st)
    out = []
    for name, fn in (("k_dwk_fwd_lds<3,1> depthwise 3x3 + BN statistics, 64ch @64x64", dw),
                     ("k_pw_fwd<64,2> 1x1 64->64 + BN statistics @64x64", pw)):
        t = timed(fn, reps=4 * NPAIR)
        out.append({"kernel": name, "bound": "hbm", "algorithmic_bytes": nbytes, "launch_us": round(t * 1e6, 2),
                    "achieved": round(nbytes / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4),
                    "note": f"cache-cold: {NPAIR} rotating buffer pairs ({NPAIR * nbytes / 1e9:.2f} GB between re-uses)"})
    return out


def measure_variant(variant, args, dev, world, rank):
    """Train-step time and forward-plan time of one model variant at batch B (synthetic inputs resident in HBM)."""
    from litehandnet_amd import get_loss, get_model, heatmap
    from litehandnet_amd.config import litehandnet_cfg
    from litehandnet_amd.train import Trainer
    cfg = litehandnet_cfg(variant)
    cfg.MODEL["ca_dropout"] = args.dropout
    torch.manual_seed(0)                       # identical random init on every rank (then broadcast anyway)
    model = get_model(cfg).to(dev).train()
    if args.sync_bn:
        from litehandnet_amd.train import prepare_model
        cfg.TRAIN["syncBN"] = True
        model = prepare_model(model, cfg)
    crit = get_loss(cfg)
    trainer = Trainer(model, crit, lr=cfg.OPTIMIZER.lr, world_size=world)
    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(1 + rank)
    img = torch.randn(B, 3, 256, 256, generator=g).to(dev)
    joints = torch.zeros(B, 21, 3)
    joints[..., :2] = torch.rand(B, 21, 2, generator=g) * 256
    target, weight = heatmap.generate_target_batch(joints.to(dev), torch.ones(B, 21, 3, device=dev), [256, 256], [64, 64], 2, True)
    meta = {"target": target, "target_weight": weight}

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(img, meta)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(img, meta)
    sync()
    el = time.perf_counter() - t0
    # max over ranks (RCCL reduces device tensors; the gloo rehearsal backend, LHN_DIST_BACKEND=gloo, takes host tensors)
    tmax = torch.tensor([el], dtype=torch.float64, device=dev if world > 1 and dist.get_backend() == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    el = float(tmax.item())
    # ---- forward-only launch duration with events on the launch stream (roofline of the forward plan run)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        model.train()
        for _ in range(3):
            model(img)
        torch.cuda.synchronize()
        reps = max(5, args.steps)
        ev0.record()
        for _ in range(reps):
            model(img)
        ev1.record()
        torch.cuda.synchronize()
        fwd_ms = ev0.elapsed_time(ev1) / reps
    del trainer, model
    torch.cuda.empty_cache()
    return el, fwd_ms


def forward_roofline(variant, B, fwd_ms, copy_gbs):
    """HBM roofline of one forward plan run (all of its launches): algorithmic bytes / measured duration; MFMA-bound models
    (hourglass: 16.7 GFLOP per image) also get the fp32 matrix-core fraction."""
    alg = ALG_FWD_BYTES[variant] * B
    traffic, stale = None, None       # measured HBM bytes per forward (rocprofv3 PMC passes, committed under profiles/)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        ent = pmc.get(f"{variant}_bs{B}_256", {})
        traffic = ent.get("traffic_bytes_per_fwd")
        stale = ent.get("src_sha16") != source_sha16()
        if stale:
            traffic = None             # kernels / plan changed since the counters were collected: do not quote them
    except (OSError, ValueError):
        pass
    ach = alg / (fwd_ms * 1e-3) / 1e9
    tf = ALG_FWD_FLOPS[variant] * B / (fwd_ms * 1e-3) / 1e12
    bound = "mfma" if tf / MFMA_FP32_PEAK_TF > ach / HBM_PEAK_GBS else "hbm"
    r = {"bound": bound, "kernel": "forward plan (one lhn_plan_run launch sequence)",
         "achieved": round(tf if bound == "mfma" else ach, 1), "peak": MFMA_FP32_PEAK_TF if bound == "mfma" else HBM_PEAK_GBS,
         "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
         "frac": round(tf / MFMA_FP32_PEAK_TF if bound == "mfma" else ach / HBM_PEAK_GBS, 4), "traffic": traffic,
         "algorithmic_bytes": alg, "launch_ms": round(fwd_ms, 4),
         "hbm_gbs": round(ach, 1), "frac_of_measured_copy": round(ach / copy_gbs, 4), "measured_copy_gbs": round(copy_gbs, 1),
         "fp32_mfma_tflops": round(tf, 2)}
    if stale is not None:
        r["traffic_stale"] = bool(stale)
    return r


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` run on its own (no WORLD_SIZE: nobody wrapped it in torch.distributed.run) starts its N
    ranks itself -- the counterpart of the reference's `mp.spawn(main, nprocs=ngpus)` (dist_train.py:264-276).  The ranks
    are fresh child processes (`python -m torch.distributed.run --standalone`-style rendezvous on 127.0.0.1) started BEFORE
    this parent makes any GPU call; the parent relays rank 0's JSON line and the children's exit code and never touches
    the GPU itself (no exec after HIP initialisation).  With fewer than N GPUs visible the ranks cannot run over RCCL (one
    GPU per rank): that is an error unless LHN_DIST_BACKEND=gloo asks for the rehearsal backend (ranks share the GPUs)."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()          # counts devices without initialising HIP
    if ndev < n and os.environ.get("LHN_DIST_BACKEND") != "gloo":
        raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible (RCCL needs one per rank; LHN_DIST_BACKEND=gloo "
                         "rehearses with shared GPUs)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--variant", default="B", choices=["A", "B", "M", "H", "L"])
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline variant only (skip variant A and the per-kernel rooflines)")
    ap.add_argument("--sync-bn", action="store_true", help="cfg.TRAIN.syncBN: SyncBatchNorm over the process group (N > 1)")
    ap.add_argument("--dropout", type=float, default=0.3, help="Dropout2d p inside channel attention (reference: 0.3)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    from litehandnet_amd.train import init_distributed

    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the two must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    B = args.batch
    el, fwd_ms = measure_variant(args.variant, args, dev, world, rank)
    extras = world == 1 and not args.no_extras
    # variant A is the model the reference registers as `litehandnet` (models/__init__.py:11): timed in the same run
    a_el, a_fwd = measure_variant("A", args, dev, world, rank) if (extras and args.variant == "B") else (None, None)
    # BASELINE config 5 (stacked hourglass, Lite-HRNet-18): same batch / steps, so that the driver's own run times them too
    others = {v: measure_variant(v, args, dev, world, rank) for v in ("H", "L")} if (extras and args.variant == "B") else {}
    if rank != 0:
        return
    copy_gbs = copy_bandwidth(dev)
    value = world * B * args.steps / el
    step_ms = el / args.steps * 1e3
    train_alg = (3 * ALG_FWD_BYTES[args.variant] + LOSS_BYTES) * B
    out = {
        "metric": "images/sec fwd+bwd @256x256 bs64 litehandnet", "value": round(value, 1), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"litehandnet variant {args.variant} ({VARIANT_NAME[args.variant]}) "
                               f"C={ {'H': 256, 'L': '40..320'}.get(args.variant, 128) }, 256x256x3 -> 21x64x64, per-GPU batch {B}, train-mode BN, fwd + TopdownHeatmapLoss + bwd + fused Adam",
                   "global_batch": world * B, "parallelism": f"dp{world}", "ca_dropout": args.dropout},
        "roofline": forward_roofline(args.variant, B, fwd_ms, copy_gbs),
        "roofline_train_step": {"bound": "hbm", "achieved": round(train_alg / (step_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(train_alg / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "forward_images_per_s": round(B / (fwd_ms * 1e-3), 1),
    }
    if a_el is not None:
        # the model the reference REGISTERS as `litehandnet` (models/__init__.py:11) is variant A; `value` is the MSRB hourglass
        # (variant B) that north_star's roofline target names -- both at the top level so that neither reads as the other
        out["value_registered_litehandnet"] = round(B * args.steps / a_el, 1)
        out["value_is"] = "variant B (MSRB hourglass, litehourglass.py); value_registered_litehandnet = variant A (liteHandNet.py)"
        a_ms = a_el / args.steps * 1e3
        out["variant_A"] = {"workload": f"litehandnet variant A ({VARIANT_NAME['A']}), same batch / steps",
                            "ms_per_step": round(a_ms, 3), "images_per_s": round(B * args.steps / a_el, 1),
                            "forward_images_per_s": round(B / (a_fwd * 1e-3), 1)}
        out["roofline_A"] = forward_roofline("A", B, a_fwd, copy_gbs)
    for v, (v_el, v_fwd) in others.items():
        out[f"variant_{v}"] = {"workload": f"{VARIANT_NAME[v]}, same batch / steps", "ms_per_step": round(v_el / args.steps * 1e3, 3),
                               "images_per_s": round(B * args.steps / v_el, 1), "forward_images_per_s": round(B / (v_fwd * 1e-3), 1)}
        out[f"roofline_{v}"] = forward_roofline(v, B, v_fwd, copy_gbs)
    if extras:
        out["roofline_kernels"] = kernel_rooflines(B, dev)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.variant)
    elif world == 1:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
    if dist.is_available() and dist.is_initialized():
        dist.barrier()                 # rank 0 is still timing its single-rank extras: tear the communicator down together
        dist.destroy_process_group()

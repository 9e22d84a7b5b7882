"""GPU parity of the training step bench.py measures (`litehandnet_amd.train.Trainer.step` = forward, TopdownHeatmapLoss,
backward, flat-gradient Adam: train/topdown_trainer.py:70-81, train/optimizer_scheduler.py:26) and of the north-star's
end-to-end accuracy criterion (PCK@0.2 of the HIP path vs the CPU reference path on the same inputs).

The multi-step trajectory is ILL-CONDITIONED by construction: Adam's first steps move every parameter by +-lr whatever
the size of its gradient, so parameters whose gradient is rounding noise take a random direction.  The reference itself,
run in fp32 and in float64 on the same data, drifts apart by 1e-3..2e-2 in the loss from step 2 on (measured when the
fixture was made, `losses` vs `losses_f64`), so the trajectory bar is that yardstick, while step 1 (loss, update
direction and size) is held to a tight bar."""
import os

import numpy as np
import pytest
import torch

from litehandnet_amd.config import litehandnet_cfg
from oracle import heatmap_np as onp
from oracle import synth, torch_ref

pytestmark = pytest.mark.gpu


def _batch(seed, n, size):
    x = synth.synth_images(n, size, seed)
    j = synth.synth_joints(n, 21, size, seed + 1)
    t = np.stack([onp.msra_generate_target(a, np.ones_like(a), [size, size], [size // 4, size // 4])[0] for a in j])
    return x, {"target": torch.from_numpy(t), "target_weight": torch.ones(n, 21, 1)}, j


def test_trainer_trajectory_golden(dev, golden_dir):
    """6 Adam steps of variant B against the REAL reference's trajectory (tests/golden/make_golden_train.py)."""
    from litehandnet_amd import get_loss, get_model
    from litehandnet_amd.train import Trainer
    g = np.load(os.path.join(golden_dir, "train_B_64.npz"))
    steps, n, size, lr = int(g["steps"]), int(g["n"]), int(g["size"]), float(g["lr"])
    cfg = litehandnet_cfg("B")
    cfg.MODEL["ca_dropout"] = 0.0
    ours = get_model(cfg)
    ref64 = torch_ref.get_model(cfg, p_drop=0.0)
    sd = synth.synth_state_dict(ref64, int(g["weights_seed"]))
    ref64.load_state_dict(sd)
    ref64 = ref64.double().train()
    ours.load_state_dict(sd)
    ours.to(dev).train()
    trainer = Trainer(ours, get_loss(cfg), lr=lr, world_size=1)
    n0 = float(trainer.fp.flat.double().norm())
    assert abs(n0 - float(g["param_norm_before"])) <= 1e-6 * n0

    # ---- step 1: loss, and the Adam update against the float64 gradient of the oracle
    x, meta, _ = _batch(50, n, size)
    before = {k: p.detach().clone() for k, p in ours.named_parameters()}
    loss = trainer.step(x.to(dev), {k: v.to(dev) for k, v in meta.items()})
    losses = [float(loss.detach())]
    assert abs(losses[0] - float(g["losses"][0])) <= 1e-5 * abs(float(g["losses"][0])), (losses[0], g["losses"][0])
    y = ref64(x.double())
    l64, _ = torch_ref.TopdownHeatmapLoss(cfg)(y, {k: v.double() for k, v in meta.items()})
    l64.backward()
    tot = agree = sized = 0
    for k, p in ours.named_parameters():
        g64 = dict(ref64.named_parameters())[k].grad
        d = (p.detach() - before[k]).cpu().double()
        clear = g64.abs() > torch.clamp(0.1 * g64.abs().mean(), min=2e-6)    # well above rounding noise and Adam's eps
        if clear.sum() == 0:
            continue
        tot += int(clear.sum())
        agree += int((torch.sign(-d[clear]) == torch.sign(g64[clear])).sum())
        # first Adam step: |update| = lr * |g| / (|g| + eps) ~= lr
        sized += int(((d[clear].abs() - lr).abs() <= 0.02 * lr).sum())
    # a handful of elements sit where fp32 rounding decides the sign / size of a near-cancelling gradient sum: torch's own
    # fp32 CPU run of the reference agrees with float64 on 99.73 % of these elements (99.02 % at a 0.01 x mean threshold)
    assert tot > 100000 and agree >= 0.995 * tot and sized >= 0.995 * tot, (agree, sized, tot)

    # ---- steps 2..: the reference's own fp32-vs-float64 drift is the yardstick
    for s in range(1, steps):
        x, meta, _ = _batch(50 + 2 * s, n, size)
        losses.append(float(trainer.step(x.to(dev), {k: v.to(dev) for k, v in meta.items()}).detach()))
    ref, f64 = g["losses"], g["losses_f64"]
    for s in range(1, steps):
        bar = max(5 * abs(ref[s] - f64[s]), 2e-2 * abs(ref[s]))
        assert abs(losses[s] - ref[s]) <= bar, (s, losses, ref.tolist())
    assert losses[-1] < losses[0]                                    # and it trains
    d = float((trainer.fp.flat.double().norm()))
    assert abs(d - float(g["param_norm_after"])) <= 1e-3 * d
    upd = torch.cat([(p.detach() - before[k]).reshape(-1) for k, p in ours.named_parameters()]).double().norm()
    assert abs(float(upd) - float(g["update_norm"])) <= 0.1 * float(g["update_norm"])


def test_pck_parity_end_to_end(dev):
    """North star: PCK@0.2 of the HIP path within 0.1 % of the CPU reference path.  Same weights and images through
    (HIP backbone -> HIP decode -> HIP PCK) and (oracle backbone -> numpy decode -> numpy PCK) at 256x256; the ground
    truth is the float64 oracle's own prediction plus noise sized so that about half of the joints pass the threshold."""
    from litehandnet_amd import get_model, heatmap
    n, size = 16, 256
    cfg = litehandnet_cfg("B")
    cfg.MODEL["ca_dropout"] = 0.0
    ours, ref = get_model(cfg), torch_ref.get_model(cfg, p_drop=0.0)
    sd = synth.synth_state_dict(ref, 61)
    ref.load_state_dict(sd)
    ours.load_state_dict(sd)
    ours.to(dev).train()
    ref.train()
    x = synth.synth_images(n, size, 62)
    with torch.no_grad():
        hr = ref(x).numpy()
        hg = ours(x.to(dev))
    # fp32 HIP vs fp32 CPU (each ~1e-4 of the map's range away from float64; the arbitrated bar is in test_model_gpu.py)
    assert np.abs(hg.cpu().numpy() - hr).max() <= 1e-3 * np.abs(hr).max()
    r = np.random.Generator(np.random.PCG64(63))
    center = r.uniform(100, 156, (n, 2)).astype(np.float32)
    scale = r.uniform(0.8, 1.4, (n, 2)).astype(np.float32)
    _, pr, _ = onp.keypoints_from_heatmaps(hr, center, scale, "default")
    _, pg, _ = heatmap.keypoints_from_heatmaps(hg, center, scale, post_process="default")
    pg = pg.cpu().numpy()
    same = np.all(pg == pr, axis=2).mean()
    assert same >= 0.99, same                                         # argmax + quarter-pixel shift + back-transform
    bbox = (200.0 * scale.max(1, keepdims=True)).astype(np.float32)
    norm = np.concatenate([bbox, bbox], 1)
    ang, rad = r.uniform(0, 2 * np.pi, (n, 21)), r.uniform(0.0, 0.4, (n, 21)) * bbox
    gt = (pr + np.stack([rad * np.cos(ang), rad * np.sin(ang)], 2)).astype(np.float32)
    mask = np.ones((n, 21), bool)
    _, pck_ref, cnt_ref = onp.keypoint_pck_accuracy(pr.copy(), gt.copy(), mask.copy(), 0.2, norm.copy())
    _, pck_hip, cnt_hip = heatmap.keypoint_pck_accuracy(pg, gt, mask, 0.2, norm)
    assert cnt_ref == cnt_hip == 21
    assert 0.3 < pck_ref < 0.7                                        # a non-trivial operating point
    assert abs(pck_hip - pck_ref) <= 1e-3, (pck_hip, pck_ref)


def test_ddp_wrapper_single_rank(dev):
    """Boundary row (b): the model must survive `DistributedDataParallel(model, device_ids=[gpu], find_unused_parameters=...)`
    (train/spawn_dist.py:49-52) with the reference's own loop and optimizer.  With a process group present the engine
    hands every parameter's gradient to autograd, so DDP's reducer hooks fire; here: one rank over RCCL, gradients equal
    to the direct (flat-buffer) path, and a second iteration runs (DDP raises if a reduction never finished)."""
    import socket

    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    from litehandnet_amd import get_loss, get_model
    from litehandnet_amd.engine import Engine
    cfg = litehandnet_cfg("B")
    cfg.MODEL["ca_dropout"] = 0.0
    a, b = get_model(cfg), get_model(cfg)
    sd = synth.synth_state_dict(torch_ref.get_model(cfg, p_drop=0.0), 71)
    a.load_state_dict(sd)
    b.load_state_dict(sd)
    a.to(dev).train()
    b.to(dev).train()
    crit = get_loss(cfg)
    x, meta, _ = _batch(72, 8, 64)
    x, meta = x.to(dev), {k: v.to(dev) for k, v in meta.items()}
    ea = Engine(a)
    ea.grads_via_autograd = False
    a.__dict__["_engine"] = ea
    la, _ = crit(a(x), meta)
    la.backward()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        ddp = DDP(b, device_ids=[dev.index], find_unused_parameters=True)
        opt = torch.optim.Adam(b.parameters(), lr=5e-4)
        lb, _ = crit(ddp(x), meta)
        opt.zero_grad()
        lb.backward()
        assert abs(float(la.detach()) - float(lb.detach())) <= 1e-6 * abs(float(la.detach()))
        gmax = max(float(p.grad.norm()) for p in a.parameters())
        for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            assert pb.grad is not None, k
            assert float((pa.grad - pb.grad).norm()) <= 1e-3 * float(pa.grad.norm()) + 1e-5 * gmax, k
        before = [p.detach().clone() for p in b.parameters()]
        opt.step()
        assert any(not torch.equal(p0, p) for p0, p in zip(before, b.parameters()))
        l2, _ = crit(ddp(x), meta)                      # second iteration through the wrapper
        opt.zero_grad()
        l2.backward()
        assert torch.isfinite(l2.detach()) and all(torch.isfinite(p.grad).all() for p in b.parameters())
    finally:
        dist.destroy_process_group()


def _dp_worker(rank, world, port, outdir):
    """One data-parallel rank (train/spawn_dist.py:10-52): SyncBatchNorm over the process group + gradient all-reduce.
    Both ranks share cuda:0 and talk over gloo (the rehearsal backend; RCCL needs one GPU per rank)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist

    from litehandnet_amd import get_model
    from litehandnet_amd.engine import Engine
    from litehandnet_amd.train import allreduce_mean_, prepare_model
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        cfg = litehandnet_cfg("B")
        cfg.MODEL["ca_dropout"] = 0.0
        cfg.TRAIN["syncBN"] = True
        model = get_model(cfg)
        model.load_state_dict(synth.synth_state_dict(torch_ref.get_model(cfg, p_drop=0.0), 80))
        model = prepare_model(model.to(dev).train(), cfg)
        assert any(isinstance(m, torch.nn.SyncBatchNorm) for m in model.modules())
        eng = Engine(model)
        eng.grads_via_autograd = False
        model.__dict__["_engine"] = eng
        n = 8 // world
        x = synth.synth_images(8, 64, 81)[rank * n:(rank + 1) * n].to(dev)
        g = torch.from_numpy(np.random.Generator(np.random.PCG64(82)).standard_normal((8, 21, 16, 16)).astype(np.float32))
        y = model(x)
        ((y * g[rank * n:(rank + 1) * n].to(dev)).sum() / n).backward()
        fg = allreduce_mean_(eng.flat_grads)
        # the replicated [32][2][C] sums are folded on the device before the exchange: [2][C] doubles per BatchNorm on the wire
        plan = next(iter(eng.plans.values()))
        wire = sum((n // 32 if rep else n) for _, _, n, rep in plan.pb.sync_points[1])
        assert plan.sync_wire_doubles == wire and wire * 32 <= sum(n for _, _, n, _ in plan.pb.sync_points[1]) + 31 * 8 * 2 * 128
        sd = model.state_dict()
        k = sorted(k for k in sd if k.endswith("running_var"))[3]
        np.savez(os.path.join(outdir, f"r{rank}.npz"), y=y.detach().cpu().numpy(), grads=fg.cpu().numpy(), rv=sd[k].cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_syncbn_data_parallel(dev, tmp_path):
    """Two real processes (gloo, both on cuda:0), cfg.TRAIN.syncBN=True, half a batch each: heatmaps, all-reduced flat
    gradient and running statistics equal plain BatchNorm over the whole batch as computed by the float64 oracle on the CPU."""
    import socket

    import torch.multiprocessing as mp

    from litehandnet_amd import get_model
    from litehandnet_amd.engine import Engine
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # arbiter: the float64 ORACLE on the whole batch with plain BatchNorm (not this library); yardstick: its fp32 CPU run
    cfg = litehandnet_cfg("B")
    cfg.MODEL["ca_dropout"] = 0.0
    ref = torch_ref.get_model(cfg, p_drop=0.0)
    ref.load_state_dict(synth.synth_state_dict(ref, 80))
    ref.train()
    x = synth.synth_images(8, 64, 81)
    g = torch.from_numpy(np.random.Generator(np.random.PCG64(82)).standard_normal((8, 21, 16, 16)).astype(np.float32))
    import copy
    ref32 = copy.deepcopy(ref)
    y32 = ref32(x)
    ((y32 * g).sum() / 8).backward()
    ref = ref.double()
    y = ref(x.double())
    ((y * g.double()).sum() / 8).backward()
    ref_y = y.detach().numpy()
    sd = ref.state_dict()
    k = sorted(k for k in sd if k.endswith("running_var"))[3]
    r = [np.load(os.path.join(str(tmp_path), f"r{i}.npz")) for i in range(2)]
    ys = np.concatenate([r[0]["y"], r[1]["y"]])
    e32y = np.abs(y32.detach().numpy() - ref_y).max() / np.abs(ref_y).max()
    assert np.abs(ys - ref_y).max() <= max(1e-4, 3 * e32y) * np.abs(ref_y).max()
    assert np.array_equal(r[0]["grads"], r[1]["grads"])                  # every rank holds the same reduced gradient
    p32 = dict(ref32.named_parameters())
    gnorm = float(sum(float(p.grad.norm()) ** 2 for p in ref.parameters()) ** 0.5)
    errs, e32s, off = {}, {}, 0
    for kk, p in ref.named_parameters():                                # flat layout: tensors padded to 4 floats
        a, b = r[0]["grads"][off:off + p.numel()].astype(np.float64), p.grad.numpy().reshape(-1)
        den = np.linalg.norm(b) + 1e-3 * gnorm / 20
        errs[kk] = float(np.linalg.norm(a - b) / den)
        e32s[kk] = float(np.linalg.norm(p32[kk].grad.numpy().reshape(-1) - b) / den)
        off += (p.numel() + 3) // 4 * 4
    worst32 = max(e32s.values())
    # (N = 8, BatchNorm over 8 attention values: single gradients are ill-conditioned, see test_odd_batches; the bar is
    # 4x the oracle's own fp32 error on that tensor or twice its worst tensor)
    bad = {kk: (e, e32s[kk]) for kk, e in errs.items() if e >= max(5e-3, 4 * e32s[kk], 2 * worst32)}
    assert not bad, sorted(bad.items(), key=lambda t: -t[1][0])[:5]
    assert np.allclose(r[0]["rv"], sd[k].numpy(), rtol=1e-4, atol=1e-6)
    assert np.array_equal(r[0]["rv"], r[1]["rv"])


def test_gradient_accumulation_semantics(dev):
    """torch semantics outside the Trainer: backward without zero_grad accumulates into param.grad, zero_grad resets."""
    from litehandnet_amd import litehourglass as lh
    m = lh.MSRB(64, 64, "ca", p_drop=0.0)
    m.load_state_dict(synth.synth_state_dict(torch_ref.MSRB(64, 64, "ca", 0.0), 91))
    m.to(dev).train()
    x = torch.randn(4, 64, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
    g = torch.randn(4, 64, 16, 16, generator=torch.Generator().manual_seed(6)).to(dev)

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-12))
    m(x).backward(g)
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    # BatchNorm running statistics move between the passes, the batch-statistics gradients do not depend on them
    m(x).backward(g)
    for k, p in m.named_parameters():
        assert rel(p.grad, 2 * g1[k]) < 1e-4, k
    m.zero_grad()
    m(x).backward(g)
    for k, p in m.named_parameters():
        assert rel(p.grad, g1[k]) < 1e-4, k
    for p in m.parameters():                       # in-place zeroing keeps the published views: still a fresh gradient
        p.grad.zero_()
    m(x).backward(g)
    for k, p in m.named_parameters():
        assert rel(p.grad, g1[k]) < 1e-4, k


def test_synthetic_training_learns(dev):
    """End to end on the GPU: images of 21 colour-coded blobs -> backbone -> loss -> backward -> Adam, device-side target
    encode / DARK decode / PCK (scripts/train_synthetic.py).  300 steps take PCK@0.2 from chance (~0.04) past 0.35
    (measured 0.55; 0.91 after 1500 steps)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "train_synthetic.py"), "--steps", "300"],
                         capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("step ")]
    first, last = float(lines[0].split("PCK@0.2")[1].split()[0]), float(lines[-1].split("PCK@0.2")[1].split()[0])
    assert first < 0.1 and last > 0.35, (first, last)


def test_bench_two_ranks_gloo_rehearsal(dev):
    """PLAIN `python bench.py --gpus 2` (the form of the driver's N = 1 command; no torch.distributed.run around it): bench.py
    starts its two ranks itself before touching the GPU (dist_train.py:264-276 `mp.spawn`), here with the gloo rehearsal
    backend (both ranks on cuda:0; RCCL needs one GPU per rank) -- rendezvous, parameter broadcast, side-stream gradient
    exchange, barrier + max-over-ranks timing, ONE JSON line from rank 0 with n_gpus 2 and global batch 128."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(LHN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    assert d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 128 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-2 * d["value"]      # whole-job images / max-over-ranks time
    # without the rehearsal switch two ranks on a one-GPU box must fail loudly, not measure one GPU
    if torch.cuda.device_count() < 2:
        env.pop("LHN_DIST_BACKEND")
        bad = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120, cwd=root)
        assert bad.returncode != 0 and "GPU" in bad.stderr


def test_run_to_run_spread(dev):
    """DEFAULT mode (the deterministic one is LHN_DETERMINISTIC=1, see test_deterministic_mode_is_bit_reproducible): BatchNorm
    statistics, weight-gradient replicas and the gate / pool reductions are accumulated with double / float atomics, so the
    last bits depend on workgroup arrival order.  This bounds the spread: the SAME training forward + backward (same weights, inputs, dropout masks) run three times gives outputs and
    parameter gradients equal to 1e-5 relative (norm-wise), running statistics to 1e-6."""
    from litehandnet_amd import get_loss, get_model
    cfg = litehandnet_cfg("B")
    cfg.MODEL["ca_dropout"] = 0.0
    crit = get_loss(cfg)
    x, meta, _ = _batch(41, 8, 128)
    x = x.to(dev)
    meta = {k: v.to(dev) for k, v in meta.items()}
    sd0 = None
    runs = []
    for _ in range(3):
        m = get_model(cfg)
        if sd0 is None:
            sd0 = {k: v.clone() for k, v in synth.synth_state_dict(m, 23).items()}
        m.load_state_dict(sd0)
        m.to(dev).train()
        y = m(x)
        loss, _ = crit(y, meta)
        loss.backward()
        runs.append((y.detach().double(), {k: p.grad.double() for k, p in m.named_parameters()},
                     {k: b.double() for k, b in m.named_buffers() if "running" in k}, float(loss)))

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-30))
    y0, g0, b0, l0 = runs[0]
    gn = max(float(v.norm()) for v in g0.values())
    for y, g, b, l in runs[1:]:
        assert rel(y, y0) < 1e-5 and abs(l - l0) <= 1e-6 * abs(l0)
        for k in g0:
            # (parameters whose true gradient is zero -- a bias in front of a BatchNorm -- hold pure rounding noise: floor)
            assert float((g[k] - g0[k]).norm()) <= 1e-5 * (float(g0[k].norm()) + 1e-2 * gn), k
        for k in b0:
            assert rel(b[k], b0[k]) < 1e-6, k




def test_deterministic_mode_is_bit_reproducible(dev):
    """LHN_DETERMINISTIC=1 (include/lhn.h: lhn_deterministic): two runs of the same two training steps -- outputs, loss,
    every parameter gradient, every running statistic, dropout on -- agree BIT FOR BIT for variants B, A, mynet, Lite-HRNet
    and the stacked hourglass (tests/check_determinism.py; without the switch the same script reports a mismatch for every one).
    The switch is read once per process, so the check runs in a child process (started before it touches the GPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LHN_DETERMINISTIC="1", LHN_REPO=root)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "check_determinism.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "DET" in r.stdout and "False" not in r.stdout.split("DET")[1]


def test_fused_finalize_agrees_with_separate_finalize(dev, tmp_path):
    """LHN_FUSE_FINALIZE=1 (include/lhn.h: lhn_bnfin -- the last workgroup of a convolution writes the BatchNorm table; two-level
    arrival tickets, block-wide replica fold) against the default separate finalize launch: two training steps of variant B,
    outputs, loss, every gradient and every running statistic.  Same sums, another fold order: fp32-level agreement.  The
    switch is read once per process, so both runs are child processes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for mode in ("0", "1"):
        out = str(tmp_path / f"fin{mode}.npz")
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "finalize_child.py"), out], env=dict(os.environ, LHN_FUSE_FINALIZE=mode),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
        res.append(np.load(out))
    a, b = res
    assert set(a.files) == set(b.files)
    gmax = max(float(np.linalg.norm(a[k])) for k in a.files if k.startswith("g0."))
    for k in a.files:
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        if k.startswith("g"):
            assert np.linalg.norm(x - y) <= 1e-4 * (np.linalg.norm(x) + 1e-3 * gmax), k
        else:
            assert np.abs(x - y).max() <= 1e-5 * (np.abs(x).max() + 1e-12), k


def test_flat_adam_matches_torch_adam(dev):
    """lhn_adam_step / train.FlatAdam against torch.optim.Adam on the CPU (the reference's optimizer, dist_train.py:64-69): eight
    steps on seeded gradients, parameters and both moments within fp32 rounding; the state dictionaries interchange; weight decay
    and a length that is not a multiple of 4 included."""
    from litehandnet_amd.train import FlatAdam
    for n, wd in ((289_813, 0.0), (1001, 1e-2)):
        g0 = torch.Generator().manual_seed(n)
        p_ref = torch.nn.Parameter(torch.randn(n, generator=g0))
        p_our = torch.nn.Parameter(p_ref.detach().clone().to(dev))
        ref = torch.optim.Adam([p_ref], lr=5e-4, weight_decay=wd)
        our = FlatAdam([p_our], lr=5e-4, weight_decay=wd)
        for step in range(8):
            g = torch.randn(n, generator=g0) * (10.0 ** (step % 3 - 1))
            p_ref.grad, p_our.grad = g.clone(), g.clone().to(dev)
            ref.step(); our.step()
            if step == 3:      # state written by torch's Adam continues in ours and vice versa
                sd_r, sd_o = ref.state_dict(), our.state_dict()
                assert set(sd_r["state"][0]) == set(sd_o["state"][0]) and float(sd_r["state"][0]["step"]) == float(sd_o["state"][0]["step"]) == 4
                our.load_state_dict({"state": {0: {k: (v.clone().to(dev) if k != "step" else v.clone()) for k, v in sd_r["state"][0].items()}},
                                     "param_groups": sd_o["param_groups"]})
                p_our.data.copy_(p_ref.data)
        sr, so = ref.state[p_ref], our.state[p_our]
        for a, b, name in ((p_ref.data, p_our.data, "param"), (sr["exp_avg"], so["exp_avg"], "exp_avg"), (sr["exp_avg_sq"], so["exp_avg_sq"], "exp_avg_sq")):
            e = float((a.double() - b.cpu().double()).abs().max() / a.double().abs().max())
            assert e < 2e-6, (n, name, e)
        assert float(so["step"]) == 8


"""CPU, world_size 2, gloo: the data-parallel pieces of litehandnet_amd.train (flat parameter buffer broadcast +
one all-reduce(SUM)/world of the flat gradient = DistributedDataParallel semantics, train/spawn_dist.py:49-52)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from litehandnet_amd.train import FlatParams, allreduce_mean_


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # ranks start with DIFFERENT weights
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3, padding=1), torch.nn.Conv2d(5, 2, 1))
    fp = FlatParams(model)
    fp.broadcast(0)                                     # ... and agree after the broadcast of the flat buffer
    flat0 = fp.flat.clone()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 3, 6, 6, generator=g)
    y = torch.randn(8, 2, 6, 6, generator=g)
    xs, ys = x[rank::world], y[rank::world]            # shard the batch
    loss = ((model(xs) - ys) ** 2).mean()
    loss.backward()
    # flat gradient in the FlatParams layout (each tensor padded to a multiple of 4 floats)
    flat_g = torch.zeros_like(fp.flat)
    off = 0
    for p in fp.params:
        flat_g[off:off + p.numel()] = p.grad.reshape(-1)
        off += (p.numel() + 3) // 4 * 4
    allreduce_mean_(flat_g)
    # reference: the full batch on one process
    torch.manual_seed(100)
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3, padding=1), torch.nn.Conv2d(5, 2, 1))
    lref = ((ref(x) - y) ** 2).mean()
    lref.backward()
    gref = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 4)) for p in ref.parameters()])
    pref = torch.cat([torch.nn.functional.pad(p.data.reshape(-1), (0, (-p.numel()) % 4)) for p in ref.parameters()])
    ok = torch.allclose(flat0, pref) and torch.allclose(flat_g, gref, rtol=1e-5, atol=1e-7)
    # the parameters are views of the flat buffer: an update of the single leaf moves every tensor
    fp.leaf.data.add_(1.0)
    ok = ok and all(torch.allclose(p.data.reshape(-1), fp.flat[o:o + p.numel()]) for p, o in zip(fp.params, _offsets(fp.params)))
    out[rank] = bool(ok)
    dist.destroy_process_group()


def _offsets(params):
    off, res = 0, []
    for p in params:
        res.append(off)
        off += (p.numel() + 3) // 4 * 4
    return res


def test_flat_allreduce_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] and out[1]


def test_gradient_handoff_follows_the_process_group():
    """Engine.grads_via_autograd = None (automatic): per-parameter gradients go through autograd -- so that the hooks of a
    DistributedDataParallel wrapper fire -- exactly while a torch.distributed process group exists; an explicit setting wins.
    Also: SyncBatchNorm mode needs converted modules AND more than one rank."""
    from litehandnet_amd import get_model
    from litehandnet_amd.config import litehandnet_cfg
    from litehandnet_amd.engine import Engine
    from litehandnet_amd.train import all_reduce_sum_, broadcast_, prepare_model
    cfg = litehandnet_cfg("B")
    m = get_model(cfg)
    eng = Engine(m)
    assert eng._via_autograd() is False
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert eng._via_autograd() is True
        eng.grads_via_autograd = False                   # what Trainer does
        assert eng._via_autograd() is False
        eng.grads_via_autograd = None
        cfg.TRAIN["syncBN"] = True
        assert prepare_model(m, cfg) is m                # one rank: nothing to synchronise, modules untouched
        assert eng.sync_config() is None
        t = torch.arange(4.0)
        assert torch.equal(all_reduce_sum_(t.clone()), t) and torch.equal(broadcast_(t.clone(), 0), t)
    finally:
        dist.destroy_process_group()
    assert eng._via_autograd() is False


def test_bench_gpus_must_match_world_size():
    """`bench.py --gpus N` never reports a line for another rank count: a WORLD_SIZE that disagrees with --gpus is an error
    (and with no WORLD_SIZE at all bench.py starts its own N ranks: tests/test_train_gpu.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr

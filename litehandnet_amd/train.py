"""Training-step harness: the build's counterpart of train/topdown_trainer.py:68-87 (train_one_epoch body),
train/spawn_dist.py:10-66 (process group + DDP wrap) and dist_train.py:64-69 (Adam, lr * world_size).

MI355X-first differences: parameters and gradients live in two flat fp32 buffers, the optimizer is ONE fused
Adam over the flat parameter, and data parallelism is ONE RCCL all-reduce of the flat gradient (SUM, then
divide by world) instead of DDP's bucketed reducer -- same arithmetic as DistributedDataParallel.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist
from torch import nn

from . import _lib
from .engine import Engine
from .plan import invalidate_tables


def init_distributed(backend=None):
    """One process per GPU (train/spawn_dist.py:10-32).  Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or os.environ.get("LHN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def _staged(t, group=None):
    """RCCL ("nccl") reduces device tensors in place.  The gloo backend is the REHEARSAL path (several ranks sharing one
    GPU, or no GPU at all): device tensors are staged through the host for it."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum_(t, group=None):
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def broadcast_(t, src=0, group=None):
    if _staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src, group=group)
    return t


def prepare_model(model, cfg):
    """train/spawn_dist.py:37-38: `cfg.TRAIN.syncBN` converts every BatchNorm to nn.SyncBatchNorm (parameter containers
    here; the engine sees them and runs the plan in SyncBatchNorm mode: statistics all-reduced over the process group
    between every convolution and its finalize -- `Engine.sync_config`)."""
    if cfg.TRAIN.get("syncBN", False) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    return model


class FlatParams:
    """Re-homes every parameter of `model` into one flat buffer (views stay valid nn.Parameters, so state_dict
    keys / shapes are untouched) and exposes a single leaf whose .grad is the engine's flat gradient."""

    def __init__(self, model):
        self.params = [p for p in model.parameters()]
        # same layout as Engine.flat_grads: every tensor padded to a multiple of 4 floats (16-byte aligned)
        n = sum((p.numel() + 3) // 4 * 4 for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            off += (k + 3) // 4 * 4
        self.leaf = nn.Parameter(self.flat, requires_grad=True)   # aliases the same storage

    def broadcast(self, src=0):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            broadcast_(self.flat, src)
            invalidate_tables()


def allreduce_mean_(flat_grads, group=None):
    """DDP semantics (train/spawn_dist.py:49-52): all-reduce(SUM) of the gradients, divided by the world size.
    One collective over the flat buffer -> a single RCCL call over xGMI."""
    if dist.is_available() and dist.is_initialized():
        w = dist.get_world_size(group)
        if w > 1:
            all_reduce_sum_(flat_grads, group)
            flat_grads.div_(w)
    return flat_grads


class FlatAdam(torch.optim.Adam):
    """torch.optim.Adam whose step() is ONE launch of lhn_adam_step per (flat, contiguous fp32, GPU) parameter.  The state layout is
    torch's own ({"step": tensor, "exp_avg", "exp_avg_sq"} per parameter, `step` a CPU scalar tensor as in torch's default form),
    so `state_dict()` / `load_state_dict()` interchange with the reference's `torch.optim.Adam` (dist_train.py:64-69).
    amsgrad / maximize / complex parameters are not supported (the reference uses none)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False, fused=False)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize"):
                raise _lib.LhnError("FlatAdam: amsgrad / maximize are not built")
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous() and g.dtype == torch.float32):
                    raise _lib.LhnError("FlatAdam: parameters and gradients must be contiguous fp32 GPU tensors")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                lr = group["lr"]
                _lib.check(L.lhn_adam_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(st["exp_avg"]), _lib.ptr(st["exp_avg_sq"]),
                                           C.c_int64(p.numel()), C.c_double(float(lr)), C.c_double(float(b1)), C.c_double(float(b2)),
                                           C.c_double(float(group["eps"])), C.c_double(float(group["weight_decay"])),
                                           C.c_int64(int(st["step"].item())), _lib.stream()), "lhn_adam_step")
        return loss


class Trainer:
    """step(img, meta): forward -> criterion -> zero_grad -> backward -> [all-reduce] -> Adam step
    (topdown_trainer.py:70-81).  Loss values stay on the device (no per-step .item())."""

    def __init__(self, model, criterion, lr=5e-4, world_size=1, optimizer="Adam"):
        _lib.require_device()
        self.model, self.criterion = model, criterion
        eng = model.__dict__.get("_engine")
        if eng is None:
            eng = Engine(model)
            model.__dict__["_engine"] = eng
        self.engine = eng
        eng.grads_via_autograd = False                 # this class owns the flat gradient and its all-reduce
        eng.accumulate_published = False               # ... and overwrites it every step (zero_grad of the flat leaf)
        self.fp = FlatParams(model)
        self.fp.broadcast(0)
        for b in model.buffers():                      # BN running statistics start identical on every rank
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and b.is_floating_point():
                broadcast_(b, 0)
        self.world = world_size
        lr = lr * world_size                           # dist_train.py:68
        if optimizer.lower() == "adam":
            # (LHN_TORCH_ADAM=1: torch's fused Adam -- 41 us per step for variant B's 289 k parameters against 5 for lhn_adam_step)
            self.opt = torch.optim.Adam([self.fp.leaf], lr=lr, fused=True) if os.environ.get("LHN_TORCH_ADAM") == "1" \
                else FlatAdam([self.fp.leaf], lr=lr)
        else:
            self.opt = torch.optim.SGD([self.fp.leaf], lr=lr, momentum=0.9)
        self.loss_sum = torch.zeros((), device=self.fp.flat.device)
        # the gradient exchange runs on its own stream, ordered by events: it starts when the last weight-gradient kernel
        # (lhn_reduce_replicas) has landed and the optimizer step waits for it; whatever the compute stream still has queued
        # (loss bookkeeping, the next step's input staging) overlaps with the collective
        self.comm_stream = torch.cuda.Stream(device=self.fp.flat.device) if self._distributed() else None

    @staticmethod
    def _distributed():
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _exchange_begin(self, g):
        """Start the all-reduce of the flat gradient on the communication stream; returns the event that marks its end."""
        if self.comm_stream is None:
            return None
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ready)
            allreduce_mean_(g)
            done = torch.cuda.Event()
            done.record(self.comm_stream)
        g.record_stream(self.comm_stream)
        return done

    def step(self, img, meta):
        out = self.model(img)
        loss, _ = self.criterion(out, meta)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        g = self.engine.flat_grads
        done = self._exchange_begin(g)
        self.loss_sum += loss.detach()                 # queued behind the backward, runs while the collective is in flight
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
        self.fp.leaf.grad = g
        self.opt.step()
        invalidate_tables()            # the fused step writes the flat leaf: per-parameter version counters do not move
        return loss

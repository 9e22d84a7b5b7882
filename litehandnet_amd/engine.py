"""Execution engine: compiles a module tree into a static plan per input shape and runs it through
`lhn_plan_run` behind one `torch.autograd.Function` (one C call per forward / backward).

Gradients land in ONE flat fp32 buffer (views of it become `param.grad`), so data-parallel training
is a single RCCL all-reduce of that buffer (train/spawn_dist.py:49-52 uses DDP's bucketed reducer).
"""
import ctypes as C

import torch
from torch import nn

from . import _lib
from .plan import CompiledPlan, PlanBuilder


class PlanModule(nn.Module):
    """Base of every mirrored block.  `emit(pb, x, out=None)` appends the block to a plan;
    calling the module on an NCHW tensor compiles and runs a plan for this block alone."""

    def emit(self, pb, x, out=None):
        raise NotImplementedError

    def forward(self, x):
        eng = self.__dict__.get("_engine")
        if eng is None:
            eng = Engine(self)
            self.__dict__["_engine"] = eng
        return eng(x)


def _is_full_model(m):
    return getattr(m, "consumes_image", False)


class _PlanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, eng, training, *params):
        plan = eng.plan_for(x, with_backward=torch.is_grad_enabled() or ctx.needs_input_grad[0] or anchor.requires_grad)
        ctx.eng, ctx.plan, ctx.training, ctx.via_autograd = eng, plan, training, len(params) > 0
        if eng.full and ctx.needs_input_grad[0]:
            raise _lib.LhnError("the gradient with respect to the input image is not built (the stem backward computes "
                                "weight gradients only): pass the image without requires_grad")
        plan.refresh_params()
        xc = x.contiguous()
        if not eng.full:
            plan.buf_data(plan.pb.in_ref).copy_(xc.permute(0, 2, 3, 1))
        out = None
        if plan.nchw_out:
            shape = (x.shape[0], plan.pb.nchw_out_C, plan.out_hw[0], plan.out_hw[1])
            if getattr(plan, "stacked", False):        # hourglass: [N, num_stack, K, H, W] even for one stack
                shape = (shape[0], plan.pb.nchw_stacks) + shape[1:]
            out = torch.empty(shape, dtype=torch.float32, device=x.device)
        ctx.sync = eng.sync_config() if training else None
        plan.run(0, xc if eng.full else None, out, training, sync=ctx.sync)
        ctx.serial = plan.fwd_serial
        if eng.full:
            ctx.save_for_backward(xc)
        if not plan.nchw_out:
            out = plan.buf_data(plan.pb.out_ref).permute(0, 3, 1, 2).contiguous()
        return out

    @staticmethod
    def backward(ctx, dout):
        eng, plan = ctx.eng, ctx.plan
        if not ctx.training:
            raise _lib.LhnError("backward through an eval-mode (running-statistics) plan is not supported")
        if plan.n_bwd == 0:
            raise _lib.LhnError("plan was compiled without a backward pass")
        # One workspace per (module, input shape): the activations a backward reads are those of the LAST forward of that
        # shape, and the backward kernels overwrite gradients in place.  Anything else must fail loudly, not silently differ
        # from torch: two forwards then two backwards, a grad-enabled forward in between, backward(retain_graph=True) twice.
        if ctx.serial != plan.fwd_serial:
            raise _lib.LhnError("backward of a stale forward: another forward of the same input shape ran on this module "
                                "since (one workspace per shape; call backward before the next forward of that shape, or "
                                "run the extra forward under torch.no_grad())")
        if plan.bwd_serial == ctx.serial:
            raise _lib.LhnError("second backward through the same forward: the first one consumed the saved activations "
                                "(retain_graph is not supported; run the forward again)")
        plan.bwd_serial = ctx.serial
        # torch semantics: a backward without a zero_grad in between ACCUMULATES.  Gradients published by the previous
        # backward are views of the flat buffer this one overwrites, so keep them aside first (rare path: one 1-9 MB copy).
        held = eng.flat_grads.clone() if (not ctx.via_autograd and eng.accumulate_published and eng._published_live()) else None
        eng.grad_parts.zero_()
        plan.set_grads(eng.part_views)
        dx, xc, dnchw = None, None, None
        if eng.full:
            (xc,) = ctx.saved_tensors
        if plan.nchw_out:
            dnchw = dout.contiguous()
        else:
            plan.buf_data(plan.pb.out_ref, grad=True).copy_(dout.permute(0, 2, 3, 1))
        plan.run(1, xc, dnchw, True, eng.GRAD_REPLICAS, eng.grad_stride, sync=ctx.sync)
        _lib.check(_lib.lib().lhn_reduce_replicas(_lib.ptr(eng.flat_grads), _lib.ptr(eng.grad_parts),
                                                  C.c_int64(eng.grad_stride), eng.GRAD_REPLICAS, C.c_int64(eng.grad_stride),
                                                  _lib.stream()), "lhn_reduce_replicas")
        if held is not None:
            eng.flat_grads.add_(held)
        if not eng.full:
            dx = plan.buf_data(plan.pb.in_ref, grad=True).permute(0, 3, 1, 2).contiguous()
        if ctx.via_autograd:
            return (dx, None, None, None) + tuple(eng.param_grad_views)
        eng.publish_grads()
        return dx, None, None, None


class Engine:
    GRAD_REPLICAS = 16      # weight-gradient partial copies (spreads the cross-block atomic adds)

    def __init__(self, module, p_drop=None):
        self.module = module
        self.full = _is_full_model(module)
        self.plans = {}
        self.anchor = None
        self.flat_grads = None
        # True: per-parameter gradients are returned through autograd, so AccumulateGrad (and with it the hooks of
        # torch's DistributedDataParallel, train/spawn_dist.py:49-52) fires for every parameter.  False: views of the
        # flat gradient buffer are published straight into `param.grad` (no per-parameter launches; what
        # litehandnet_amd.train.Trainer uses, doing the all-reduce itself).  None = automatic: through autograd whenever
        # a torch.distributed process group exists, because the model may then sit inside a DDP wrapper.
        self.grads_via_autograd = None
        self.p_drop = p_drop
        self.sync_override = None           # (world, all_reduce_fn): tests / custom process groups
        self.mask_fn = None                 # callable(plan) filling plan.mask_slices: injected dropout masks (parity tests)
        # direct mode: add to gradients that are still published in `param.grad` (no zero_grad since the last backward).
        # litehandnet_amd.train.Trainer owns the flat buffer and its zeroing, and switches this off.
        self.accumulate_published = True

    def sync_config(self):
        """SyncBatchNorm (train/spawn_dist.py:37-38, cfg.TRAIN.syncBN): active when the model holds nn.SyncBatchNorm
        modules (nn.SyncBatchNorm.convert_sync_batchnorm(model) as the reference does) and a process group with
        more than one rank exists.  Returns (world, all_reduce_fn) or None."""
        if self.sync_override is not None:
            return self.sync_override
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return None
        if not any(isinstance(m, nn.SyncBatchNorm) for m in self.module.modules()):
            return None
        from .train import all_reduce_sum_
        return dist.get_world_size(), all_reduce_sum_

    def _via_autograd(self):
        if self.grads_via_autograd is not None:
            return bool(self.grads_via_autograd)
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()

    # -------------------------------------------------------------- state
    def _state(self, device):
        sd = self.module.state_dict(keep_vars=True)
        tensors = list(sd.values())
        for t in tensors:
            if t.device != device:
                raise _lib.LhnError("module parameters must live on the input's GPU (call .cuda() first)")
            if not t.is_contiguous():
                raise _lib.LhnError("parameters must be contiguous")
            if t.is_floating_point() and t.dtype != torch.float32:
                raise _lib.LhnError(f"parameters and buffers must be float32 (found {t.dtype}): the kernels compute in fp32")
        return tensors

    def _ensure_grads(self, tensors, device):
        params = [t for t in tensors if isinstance(t, nn.Parameter)]
        n = sum((p.numel() + 3) // 4 * 4 for p in params)          # every tensor starts 16-byte aligned
        if self.flat_grads is None or self.flat_grads.numel() != n or self.flat_grads.device != device:
            self.flat_grads = torch.zeros(n, dtype=torch.float32, device=device)
            self.grad_parts = torch.zeros(self.GRAD_REPLICAS * n, dtype=torch.float32, device=device)
            self.grad_stride = n
            self.anchor = torch.zeros(1, dtype=torch.float32, device=device, requires_grad=True)
        off, views, pviews, plist, parts = 0, [], [], [], []
        for t in tensors:
            if isinstance(t, nn.Parameter):
                v = self.flat_grads[off:off + t.numel()].view_as(t)
                parts.append(self.grad_parts[off:off + t.numel()])        # replica 0
                off += (t.numel() + 3) // 4 * 4
                views.append(v)
                pviews.append(v)
                plist.append(t)
            else:
                views.append(None)
                parts.append(None)
        self.grad_views, self.param_grad_views, self.param_list, self.part_views = views, pviews, plist, parts

    def _published_live(self):
        """True when some parameter's .grad is still the view this engine published (the user did not reset it)."""
        for p, g in zip(self.param_list, self.param_grad_views):
            if p.requires_grad:
                return p.grad is not None and p.grad.data_ptr() == g.data_ptr()
        return False

    def publish_grads(self):
        for p, g in zip(self.param_list, self.param_grad_views):
            if not p.requires_grad:
                continue
            if p.grad is None or p.grad.data_ptr() == g.data_ptr():
                p.grad = g
            else:
                p.grad.add_(g)

    # -------------------------------------------------------------- plans
    def plan_for(self, x, with_backward):
        _lib.require_device(x)
        if x.dim() != 4 or x.dtype != torch.float32:
            raise _lib.LhnError(f"expected a float32 NCHW tensor, got {tuple(x.shape)} {x.dtype}")
        p_drop = self.p_drop if self.p_drop is not None else getattr(self.module, "p_drop", 0.0)
        if any(getattr(m, "deploy", False) for m in self.module.modules()):
            with_backward = False           # re-parameterised (deploy) form is inference-only
        key = (tuple(x.shape), bool(with_backward), float(p_drop), x.device.index)
        plan = self.plans.get(key)
        tensors = self._state(x.device)
        if plan is not None and len(plan.state_tensors) == len(tensors) and all(
                a is b for a, b in zip(plan.state_tensors, tensors)):
            return plan
        N, Cc, H, W = x.shape
        self._ensure_grads(tensors, x.device)
        index = {id(t): j for j, t in enumerate(tensors)}
        pb = PlanBuilder(N, index, image_hw=(H, W), with_backward=with_backward, p_drop=p_drop)
        if self.full:
            if Cc != 3:
                raise _lib.LhnError("the backbone consumes a 3-channel image")
            y = self.module.emit(pb, pb.image())
        else:
            y = self.module.emit(pb, pb.input_tensor(Cc, H, W))
        nchw_out = getattr(y, "buf", None) == -2
        if not nchw_out:
            pb.set_output(y)
        plan = CompiledPlan(pb, tensors, x.device)
        plan.out_hw, plan.nchw_out = (y.H, y.W), nchw_out
        plan.stacked = bool(getattr(self.module, "stacked_output", False))
        plan.engine = self              # (the plan reads `engine.mask_fn` at RUN time: setting or clearing it later takes effect)
        self.plans[key] = plan
        return plan

    def __call__(self, x):
        training = self.module.training
        if self.anchor is None or self.anchor.device != x.device:
            self._ensure_grads(self._state(x.device), x.device)
        if self._via_autograd() and torch.is_grad_enabled():
            return _PlanFn.apply(x, self.anchor, self, training, *self.param_list)
        anchor = self.anchor if torch.is_grad_enabled() else self.anchor.detach()
        return _PlanFn.apply(x, anchor, self, training)

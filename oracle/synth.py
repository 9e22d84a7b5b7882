"""ORACLE helper (test infrastructure only): deterministic weights / inputs keyed by name,
so golden fixtures never have to carry weights.  Uses numpy PCG64 only (bit-stable)."""
import zlib
import numpy as np
import torch


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), seed]))


def synth_state_dict(model, seed=0, mode="conditioned"):
    """mode 'conditioned': conv ~ N(0, 1/fan_in), BN gamma ~ U(.5,1.5), beta/mean ~ N(0,.1), var ~ U(.5,1.5).
    mode 'reference_init': what the reference's init_weights() distributions look like
    (conv ~ N(0,1); BN gamma ~ N(0,1) for variant A / 1 for B is left to the model's own init)."""
    out = {}
    bn = {n for n, m in model.named_modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)}
    for k, v in model.state_dict().items():
        r = _rng(k, seed)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
            continue
        shape = tuple(v.shape)
        if k.endswith("running_var"):
            a = r.uniform(0.5, 1.5, shape)
        elif k.endswith("running_mean"):
            a = r.normal(0, 0.1, shape)
        elif v.dim() == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            a = r.normal(0, 1.0 if mode == "reference_init" else fan_in ** -0.5, shape)
        elif k.rsplit(".", 1)[0] in bn:
            a = r.uniform(0.5, 1.5, shape) if k.endswith("weight") else r.normal(0, 0.1, shape)
        else:  # conv bias
            a = r.normal(0, 0.1, shape)
        out[k] = torch.from_numpy(np.asarray(a, np.float32))
    return out


def synth_images(n, size, seed=0):
    r = np.random.Generator(np.random.PCG64([77, seed]))
    return torch.from_numpy(r.standard_normal((n, 3, size, size)).astype(np.float32))


def synth_joints(n, k, size, seed=1, margin=0.0):
    r = np.random.Generator(np.random.PCG64([78, seed]))
    xy = r.uniform(-margin * size, (1 + margin) * size, (n, k, 2)).astype(np.float32)
    j = np.zeros((n, k, 3), np.float32)
    j[..., :2] = xy
    return j

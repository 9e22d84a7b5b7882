"""litehandnet_amd -- MI355X-native kernels for litehandnet's convolutional heatmap path.

`get_model(cfg)` / `get_loss(cfg)` mirror models/__init__.py:20-26 and loss/__init__.py:18-19 of the
reference; everything runs through liblhn.so (HIP, gfx950).  There is no CPU fallback."""
from .config import AttrDict, litehandnet_cfg  # noqa: F401


def get_model(cfg):
    from .models import get_model as _g
    return _g(cfg)


def get_loss(cfg):
    from .loss import get_loss as _g
    return _g(cfg)


def invalidate_tables():
    """See litehandnet_amd.plan.invalidate_tables: call after writing parameters / running statistics through `.data`."""
    from .plan import invalidate_tables as _i
    _i()

"""Model registry -- models/__init__.py:11-26 restricted to the hot path."""

__all__ = ["litehandnet", "litehourglass", "mynet", "hourglass", "litehrnet"]


def get_model(cfg):
    name = cfg.MODEL.name
    assert name in __all__, f"model <{name}> should be one of {__all__}"
    if name == "litehourglass":
        from .litehourglass import LiteHandNet
        return LiteHandNet(cfg)
    if name == "mynet":
        from .pose_hg_ms_att import MultiScaleAttentionHourglass
        return MultiScaleAttentionHourglass(cfg)
    if name == "hourglass":
        from .hourglassnet import HourglassNet
        return HourglassNet(cfg)
    if name == "litehrnet":
        from .lite_hrnet import LiteHRNet
        return LiteHRNet(cfg)
    from .liteHandNet import LiteHandNet
    return LiteHandNet(cfg)

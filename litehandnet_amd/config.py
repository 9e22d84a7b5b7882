"""Attribute-dict config, reading the same keys as the reference's experiment files.

The reference loads python-dict configs into `addict.Dict` (config/__init__.py:27-39);
`addict` is not a dependency here, so this is the minimal stand-in: attribute access,
`.get`, nested dicts wrapped on access.
"""


class AttrDict(dict):
    """dict with attribute access; nested dicts are converted once, so `cfg.MODEL.x = 1` sticks."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, AttrDict):
                super().__setitem__(k, AttrDict(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)


def litehandnet_cfg(variant="A", channels=128, num_joints=21, image_size=256, **model_kw):
    """Config with the keys of config/litehandnet/*_256x256_*.py (variant B keys) and
    config/litehandnet/freihand/_1_*_ca_r4_leaky.py (variant A keys)."""
    if variant == "A":
        model = dict(name="litehandnet", num_stage=4, num_block=[2, 2, 2], input_channel=channels,
                     ca_type="ca", reduction=4, activation="leakyrelu", output_channel=num_joints)
    elif variant == "B":
        model = dict(name="litehourglass", num_stage=4, msrb_ca="ca", rbu_ca="none",
                     input_channel=channels, output_channel=num_joints)
    elif variant == "M":            # `mynet`, config/mynet/_2_rhd2d_256x256_dark.py:4-11
        model = dict(name="mynet", num_stage=4, num_block=[2, 2, 2], input_channel=channels, output_channel=num_joints)
    elif variant == "H":            # stacked hourglass, config/hourglass/_2_rhd2d_256x256_dark_h2.py:4-10 (num_stack 2; _3_*_h1: 1)
        model = dict(name="hourglass", input_channel=256 if channels == 128 else channels, output_channel=num_joints,
                     num_stack=2, num_level=4)
    elif variant == "L":            # Lite-HRNet, config/litehrnet/_2_rhd2d_256x256_dark_18.py:4-9 (depth 18; _1_*_30: depth 30)
        model = dict(name="litehrnet", depth=18, output_channel=num_joints)
    else:
        raise ValueError(variant)
    model.update(model_kw)
    return AttrDict(
        MODEL=model,
        DATASET=dict(num_joints=num_joints, image_size=[image_size, image_size],
                     heatmap_size=[image_size // 4, image_size // 4]),
        PIPELINE=dict(sigma=2, kernel=(11, 11), encoding="MSRA", unbiased_encoding=True,
                      use_udp=False, simdr_split_ratio=0, target_type="GaussianHeatmap"),
        LOSS=dict(type="TopdownHeatmapLoss", loss_weight=[1.0, 0.1], auto_weight=False),
        TRAIN=dict(syncBN=False, find_unused_parameters=False, batch_per_gpu=64),
        OPTIMIZER=dict(type="Adam", lr=5e-4),
        EVAL=dict(pck_threshold=0.2),
    )

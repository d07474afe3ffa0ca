"""The architectures the reference ships (ml-mdm-matryoshka/configs/models/*.yaml) as Python
constructors, plus reduced-size variants used by the parity tests.

Numbers restate the yaml files: cc12m_64x64.yaml:42-64 (UNet-64), cc12m_256x256.yaml:34-79
(nested 64+256), cc12m_1024x1024.yaml:36-115 (nested 64+256+1024).  ``lm_dim`` is the width of
the frozen text encoder (flan-t5-xl: 2048), which the reference CLIs write into
``conditioning_feature_dim`` just before constructing the model (clis/train_parallel.py:65).
A yaml file of the reference can also be loaded directly with ``from_reference_yaml``.
"""
from .nested_unet import Nested2UNetConfig, NestedUNetConfig
from .unet import ResNetConfig, UNetConfig


def unet64_config(lm_dim: int = 2048, nesting: bool = False) -> UNetConfig:
    return UNetConfig(
        num_resnets_per_resolution=[2, 2, 2],
        attention_levels=[1, 2],
        num_attention_layers=[0, 1, 5],
        conditioning_feature_dim=lm_dim,
        conditioning_feature_proj_dim=2048,
        num_lm_head_layers=0,
        masked_cross_attention=0,
        resolution_channels=[256, 512, 768],
        skip_mid_blocks=False,
        skip_cond_emb=False,
        nesting=nesting,
        micro_conditioning="scale:64",
        resnet_config=ResNetConfig(num_groups_norm=32, dropout=0.0, use_attention_ffn=True),
    )


def nested256_config(lm_dim: int = 2048) -> NestedUNetConfig:
    return NestedUNetConfig(
        attention_levels=[],
        conditioning_feature_dim=lm_dim,
        conditioning_feature_proj_dim=-1,
        inner_config=unet64_config(-1, nesting=True),
        masked_cross_attention=1,
        micro_conditioning="scale:256",
        nesting=False,
        num_attention_layers=[0, 0, 0],
        num_resnets_per_resolution=[2, 2, 1],
        resnet_config=ResNetConfig(num_groups_norm=32, dropout=0.0, use_attention_ffn=False),
        resolution_channels=[64, 128, 256],
        skip_cond_emb=True,
        skip_inner_unet_input=False,
        skip_mid_blocks=True,
        skip_normalization=True,
        temporal_dim=1024,
    )


def nested1024_config(lm_dim: int = 2048) -> Nested2UNetConfig:
    mid = nested256_config(-1)
    mid.nesting = True
    mid.skip_normalization = False
    return Nested2UNetConfig(
        attention_levels=[],
        conditioning_feature_dim=lm_dim,
        conditioning_feature_proj_dim=-1,
        inner_config=mid,
        masked_cross_attention=1,
        micro_conditioning="scale:1024",
        nesting=False,
        num_attention_layers=[0, 0, 0],
        num_resnets_per_resolution=[2, 2, 1],
        resnet_config=ResNetConfig(num_groups_norm=32, dropout=0.0, use_attention_ffn=False),
        resolution_channels=[32, 32, 64],
        skip_cond_emb=True,
        skip_inner_unet_input=False,
        skip_mid_blocks=True,
        skip_normalization=True,
        temporal_dim=1024,
    )


# ---- reduced variants for parity tests (same topology class, minutes on CPU) ---------------
def mini_unet_config(lm_dim: int = 64, nesting: bool = False, masked: int = 0, lm_head: int = 0) -> UNetConfig:
    """2 levels (32, 256 channels), self+cross attention with FFN at level 1 (head dim 32)."""
    return UNetConfig(
        num_lm_head_layers=lm_head,
        num_resnets_per_resolution=[1, 1],
        attention_levels=[1],
        num_attention_layers=[0, 1],
        conditioning_feature_dim=lm_dim,
        conditioning_feature_proj_dim=64,
        masked_cross_attention=masked,
        resolution_channels=[32, 256],
        nesting=nesting,
        micro_conditioning="scale:16",
        temporal_dim=128,
        resnet_config=ResNetConfig(num_groups_norm=32, dropout=0.0, use_attention_ffn=True),
    )


def mini_nested_config(lm_dim: int = 64) -> NestedUNetConfig:
    """outer 32->64 conv-only U-Net at 2x resolution around ``mini_unet_config``."""
    return NestedUNetConfig(
        attention_levels=[],
        conditioning_feature_dim=lm_dim,
        conditioning_feature_proj_dim=-1,
        inner_config=mini_unet_config(-1, nesting=True),
        masked_cross_attention=1,
        micro_conditioning="scale:32",
        num_attention_layers=[0, 0],
        num_resnets_per_resolution=[1, 1],
        resnet_config=ResNetConfig(num_groups_norm=32, dropout=0.0, use_attention_ffn=False),
        resolution_channels=[32, 64],
        skip_cond_emb=True,
        skip_mid_blocks=True,
        skip_normalization=True,
        temporal_dim=128,
    )


def mini_nested2_config(lm_dim: int = 64) -> Nested2UNetConfig:
    """three levels: a 32-channel conv-only net at 4x resolution around ``mini_nested_config`` whose input is
    divided by its per-sample std (``skip_normalization=False``, as the middle net of cc12m_1024x1024.yaml:83)."""
    mid = mini_nested_config(-1)
    mid.nesting = True
    mid.skip_normalization = False
    return Nested2UNetConfig(
        attention_levels=[],
        conditioning_feature_dim=lm_dim,
        conditioning_feature_proj_dim=-1,
        inner_config=mid,
        masked_cross_attention=1,
        micro_conditioning="scale:64",
        num_attention_layers=[0, 0],
        num_resnets_per_resolution=[1, 1],
        resnet_config=ResNetConfig(num_groups_norm=32, dropout=0.0, use_attention_ffn=False),
        resolution_channels=[32, 32],
        skip_cond_emb=True,
        skip_mid_blocks=True,
        skip_normalization=True,
        temporal_dim=128,
    )


def _clean(d):
    return {k: (None if v == "None" else v) for k, v in d.items()}


def from_reference_yaml(path: str, lm_dim: int = 2048):
    """Build the model config from one of the reference's ``configs/models/*.yaml`` files."""
    import yaml

    with open(path) as f:
        doc = yaml.safe_load(f)

    def build(d):
        d = _clean(dict(d))
        rc = ResNetConfig(**d.pop("resnet_config", {}))
        inner = d.pop("inner_config", None)
        if inner is None:
            return UNetConfig(resnet_config=rc, **d)
        inner_cfg = build(inner)
        cls = Nested2UNetConfig if isinstance(inner_cfg, NestedUNetConfig) else NestedUNetConfig
        return cls(resnet_config=rc, inner_config=inner_cfg, **d)

    cfg = build(doc["unet_config"])
    cfg.conditioning_feature_dim = lm_dim
    return cfg

"""mdm_hip -- MI355X-native (gfx950) U-Net / NestedUNet denoiser hot path for ml-mdm.

Layout of the package
  _lib.py         build + ctypes binding of libmdm_hip.so (C ABI in include/mdm_hip.h)
  ops.py          torch.autograd tape entries around the C ABI
  unet.py         UNet, config dataclasses (mirror of ml_mdm.models.unet)
  nested_unet.py  NestedUNet (mirror of ml_mdm.models.nested_unet)
  diffusion.py    Diffusion / NestedDiffusion train-step + sampling call surface
  samplers.py     noise schedules + DDPM/DDIM updates
  configs.py      the three shipped architectures (cc12m 64 / 256 / 1024)
  distributed.py  one-process-per-GPU data parallel gradient reducer over RCCL
  registry.py     plug into the reference's ml_mdm.config registries when present
"""
from . import _lib  # noqa: F401
from .nested_unet import (  # noqa: F401
    Nested2UNetConfig,
    Nested3UNetConfig,
    Nested4UNetConfig,
    NestedUNet,
    NestedUNetConfig,
)
from .unet import ResNetConfig, UNet, UNetConfig  # noqa: F401

__all__ = [
    "UNet",
    "UNetConfig",
    "ResNetConfig",
    "NestedUNet",
    "NestedUNetConfig",
    "Nested2UNetConfig",
    "Nested3UNetConfig",
    "Nested4UNetConfig",
]

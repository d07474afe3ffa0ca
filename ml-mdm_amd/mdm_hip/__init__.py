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
import os as _os

# ROCclr maps a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and a queue is in order: with more
# streams than queues -- the data-parallel step has six: main, weight-gradient, communication, completion, RCCL's own, the
# null stream -- kernels of one stream wait behind barrier packets (or collectives) of another.  Measured with every bucket
# forced through RCCL in a world of one: +2.65 -> +1.6 ms per step under 8 queues, the plain step unchanged
# (tools/calls/r6/c47_hw_queues.sh).  Read when the HIP runtime initialises (the first HIP call of the process), so this
# has to happen before any; an explicit setting in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib  # noqa: F401,E402
from .nested_unet import (  # noqa: E402  # noqa: F401
    Nested2UNetConfig,
    Nested3UNetConfig,
    Nested4UNetConfig,
    NestedUNet,
    NestedUNetConfig,
)
from .unet import ResNetConfig, UNet, UNetConfig  # noqa: F401,E402

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

"""Matryoshka nesting: an outer convolutional U-Net wrapped around an inner (Nested)UNet.

Mirrors ``ml_mdm.models.nested_unet`` (reference ml-mdm-matryoshka/ml_mdm/models/nested_unet.py):
``NestedUNetConfig`` / ``Nested{2,3,4}UNetConfig`` (:21-75) and
``NestedUNet(input_channels, output_channels, config)`` (:96-160) with identical
parameter names (``inner_unet.*``, ``in_adapter.*``, ``out_adapter.*``).  The S3
``download`` helper (:78-93) is out of scope (no network): ``initialize_inner_with_pretrained``
is honoured only for a local file.

Between an outer and an inner net the feature maps stay NHWC in the compute dtype; the
NCHW fp32 convention applies only at the model boundary (images in, predictions out).
"""
import os
from dataclasses import dataclass, field

import torch
import torch.nn as nn

from . import ops
from .unet import UNet, UNetConfig, zero_module


@dataclass
class NestedUNetConfig(UNetConfig):
    inner_config: UNetConfig = field(default_factory=lambda: UNetConfig(nesting=True))
    skip_mid_blocks: bool = True
    skip_cond_emb: bool = True
    skip_inner_unet_input: bool = False
    skip_normalization: bool = False
    initialize_inner_with_pretrained: str = None
    freeze_inner_unet: bool = False
    interp_conditioning: bool = False


@dataclass
class Nested2UNetConfig(NestedUNetConfig):
    inner_config: NestedUNetConfig = field(
        default_factory=lambda: NestedUNetConfig(nesting=True, initialize_inner_with_pretrained=None)
    )


@dataclass
class Nested3UNetConfig(Nested2UNetConfig):
    inner_config: Nested2UNetConfig = field(
        default_factory=lambda: Nested2UNetConfig(nesting=True, initialize_inner_with_pretrained=None)
    )


@dataclass
class Nested4UNetConfig(Nested3UNetConfig):
    inner_config: Nested3UNetConfig = field(
        default_factory=lambda: Nested3UNetConfig(nesting=True, initialize_inner_with_pretrained=None)
    )


class NestedUNet(UNet):
    def __init__(self, input_channels, output_channels, config: NestedUNetConfig):
        super().__init__(input_channels, output_channels=output_channels, config=config)
        config.inner_config.conditioning_feature_dim = config.conditioning_feature_dim
        inner_cls = UNet if getattr(config.inner_config, "inner_config", None) is None else NestedUNet
        self.inner_unet = inner_cls(input_channels, output_channels, config.inner_config)

        c_outer, c_inner = config.resolution_channels[-1], config.inner_config.resolution_channels[0]
        if not config.skip_inner_unet_input:
            self.in_adapter = zero_module(nn.Conv2d(c_outer, c_inner, kernel_size=3, padding=1, bias=True))
        else:
            self.in_adapter = None
        self.out_adapter = zero_module(nn.Conv2d(c_inner, c_outer, kernel_size=3, padding=1, bias=True))

        self.is_temporal = [False] + list(getattr(self.inner_unet, "is_temporal", []))
        ratio = int(2 ** (len(config.resolution_channels) - 1))
        if self.inner_unet.config.nesting and self.inner_unet.model_type == "nested_unet":
            self.nest_ratio = [ratio * self.inner_unet.nest_ratio[0]] + self.inner_unet.nest_ratio
        else:
            self.nest_ratio = [ratio]

        pre = config.initialize_inner_with_pretrained
        if pre is not None and pre != "None":
            local = pre.replace("/", "_")
            for cand in (pre, local):
                if os.path.exists(cand):
                    self.inner_unet.load(cand)
                    break
            else:
                print("<-- pretrained inner checkpoint %s not found locally (no network); keeping random init -->" % pre)
        if config.freeze_inner_unet:
            for p in self.inner_unet.parameters():
                p.requires_grad = False
        if config.interp_conditioning:
            self.interp_layer1 = nn.Linear(self.temporal_dim // 4, self.temporal_dim)
            self.interp_layer2 = nn.Linear(self.temporal_dim, self.temporal_dim)

    @property
    def model_type(self):
        return "nested_unet"

    def forward_conditioning(self, *args, **kwargs):
        return self.inner_unet.forward_conditioning(*args, **kwargs)

    def forward_denoising(self, x_t, times, cond_emb=None, conditioning=None, cond_mask=None, micros={}):
        """x_t: list of NCHW fp32 images, highest resolution first (reference :168-230)."""
        temb = self._time_embedding(times, cond_emb, micros)
        if self._config.nesting:
            x_t, x_feat = x_t
        bh, bl = x_t[0].size(0), x_t[1].size(0)
        x_t_low, x_hi = x_t[1:], x_t[0]
        temb_act = self.time_states(temb[:bh] if bh != temb.shape[0] else temb)
        cond_hi = conditioning[:bh] if (conditioning is not None and bh != conditioning.shape[0]) else conditioning
        mask_hi = cond_mask[:bh] if (cond_mask is not None and bh != cond_mask.shape[0]) else cond_mask

        x = self.forward_input_layer(x_hi, normalize=not self.config.skip_normalization)
        if self._config.nesting and x_feat is not None:
            x = ops.add(x, x_feat)
        x, skips = self.forward_downsample(x, temb_act, cond_hi, mask_hi)

        x_inner = None
        if self.in_adapter is not None:
            x_inner = ops.conv(x, self.in_adapter.weight, self.in_adapter.bias)
            if bh < bl:  # mixed-resolution batch: zero features for the low-res-only samples
                x_inner = torch.cat([x_inner, x_inner.new_zeros(bl - bh, *x_inner.shape[1:])], 0)
        x_low, x_inner = self.inner_unet.forward_denoising((x_t_low, x_inner), times, cond_emb, conditioning, cond_mask, micros)
        if bh < bl:
            x_inner = x_inner[:bh]
        x = ops.conv(x_inner, self.out_adapter.weight, self.out_adapter.bias, residual=x)

        x = self.forward_upsample(x, temb_act, cond_hi, mask_hi, skips)
        x_out = self.forward_output_layer(x)
        out = [x_out] + x_low if isinstance(x_low, list) else [x_out, x_low]
        return (out, x) if self._config.nesting else out

    def print_size(self, target_image_size=256):
        pass

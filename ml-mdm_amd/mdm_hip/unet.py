"""MI355X-native U-Net denoiser: host-side mirror of the reference operator interface.

Mirrors ``ml_mdm.models.unet`` (reference file ml-mdm-matryoshka/ml_mdm/models/unet.py):
same config dataclasses (``ResNetConfig`` :44-59, ``UNetConfig`` :62-156), same
constructor signature ``UNet(input_channels, output_channels, config)`` (:580-581),
same parameter names / shapes (so checkpoints and ``state_dict``s interchange,
SURVEY.md appendix C), same ``forward(x_t, times, conditioning, cond_mask, micros)``
call surface (:971-987) with NCHW fp32 tensors in and out.

What differs is everything underneath: the ``nn.Conv2d`` / ``nn.GroupNorm`` /
``nn.Linear`` children are used purely as *parameter containers*; their
``forward`` is never called.  The arithmetic runs as NHWC kernels from
libmdm_hip (see ``ops.py``): implicit-GEMM convs on MFMA, fused
GroupNorm(+FiLM)(+SiLU), flash-style self+cross attention.  Internally every
activation is an NHWC tensor of the compute dtype (fp32 parity mode or bf16).

Not supported (raise): temporal/video mode and the LM-head ``SelfAttention1D``
blocks -- no shipped config enables them.
"""
import copy
import logging
import math
import os
from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


# --------------------------------------------------------------------------------------
# configuration (field-for-field compatible with the reference dataclasses)
# --------------------------------------------------------------------------------------
@dataclass
class ResNetConfig:
    num_channels: int = -1
    output_channels: int = -1
    num_groups_norm: int = 32
    dropout: float = 0.0
    use_attention_ffn: bool = False


def _int_list(v):
    if isinstance(v, str):
        return [int(t) for t in v.split(",") if t.strip() != ""]
    return v


@dataclass
class UNetConfig:
    num_resnets_per_resolution: str = "2"
    temporal_dim: int = None
    attention_levels: str = "2,3"
    num_attention_layers: str = "1"
    num_temporal_attention_layers: str = None
    conditioning_feature_dim: int = -1
    conditioning_feature_proj_dim: int = -1
    num_lm_head_layers: int = 0
    masked_cross_attention: int = 1
    resolution_channels: str = "128,256,256,512,1024"
    skip_mid_blocks: bool = False
    skip_cond_emb: bool = False
    nesting: bool = False
    micro_conditioning: str = None
    temporal_mode: bool = False
    temporal_spatial_ds: bool = False
    temporal_positional_encoding: bool = False
    resnet_config: ResNetConfig = field(default_factory=ResNetConfig)

    def __post_init__(self):
        self.resolution_channels = _int_list(self.resolution_channels)
        n = len(self.resolution_channels)
        self.attention_levels = _int_list(self.attention_levels) or []
        for name in ("num_attention_layers", "num_resnets_per_resolution"):
            was_str = isinstance(getattr(self, name), str)
            vals = _int_list(getattr(self, name))
            if was_str and len(vals) == 1:
                vals = vals * n
            if was_str:
                assert len(vals) == n, "%s needs one entry per resolution" % name
            setattr(self, name, vals)
        if self.num_temporal_attention_layers:
            self.num_temporal_attention_layers = _int_list(self.num_temporal_attention_layers)
        if isinstance(self.resnet_config, dict):
            self.resnet_config = ResNetConfig(**self.resnet_config)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


_SKIP_TAP = os.environ.get("MDM_HIP_SKIP_TAP", "1") != "0"   # A/B switch of the skip-gradient pass-through (ResNet.forward)


def compute_dtype() -> torch.dtype:
    """bf16 under ``torch.autocast`` (the reference's ``fp16: 1`` path, trainer.py:29-30) or
    when MDM_HIP_DTYPE=bf16; otherwise exact fp32."""
    if torch.is_autocast_enabled():
        return torch.bfloat16
    env = os.environ.get("MDM_HIP_DTYPE", "fp32").lower()
    return torch.bfloat16 if env in ("bf16", "bfloat16") else torch.float32


# --------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------
class ResNet(nn.Module):
    """GN+SiLU -> 3x3 -> GN*(1+ta)+tb -> SiLU -> 3x3 (zero-init) + shortcut   (reference unet.py:193-238)."""

    def __init__(self, time_emb_channels, config: ResNetConfig):
        super().__init__()
        self.config = config
        cin, cout = config.num_channels, config.output_channels
        self.norm1 = nn.GroupNorm(config.num_groups_norm, cin)
        self.conv1 = nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=True)
        self.time_layer = nn.Linear(time_emb_channels, cout * 2)
        self.norm2 = nn.GroupNorm(config.num_groups_norm, cout)
        self.dropout = nn.Dropout(config.dropout)
        self.conv2 = zero_module(nn.Conv2d(cout, cout, kernel_size=3, padding=1, bias=True))
        if cout != cin:
            self.conv3 = nn.Conv2d(cin, cout, kernel_size=1, bias=True)

    def forward(self, x, temb_act, tap=None):
        """x: NHWC activation; temb_act: silu(temb) [B, T] (shared by all blocks).  ``tap = (list, index)``: the entry
        ``list[index]`` holds this very x as a skip activation (reference unet.py:883-897); it is replaced by a second
        pass-through output of norm1, so that in backward the skip connection's gradient reaches x inside the
        GroupNorm kernel instead of through an accumulation kernel of the autograd engine."""
        g = self.config.num_groups_norm
        if tap is not None and x.requires_grad and tap[0][tap[1]] is x and _SKIP_TAP:
            h, x, tap[0][tap[1]] = ops.group_norm(x, self.norm1.weight, self.norm1.bias, g, self.norm1.eps, silu=True, passthrough=2)
        else:
            h, x = ops.group_norm(x, self.norm1.weight, self.norm1.bias, g, self.norm1.eps, silu=True, passthrough=True)
        h = ops.conv(h, self.conv1.weight, self.conv1.bias)
        if isinstance(temb_act, TimeStates):
            film = temb_act.film.get(id(self))
            if film is None:
                film = ops.linear(temb_act.act, self.time_layer.weight, self.time_layer.bias)
        else:
            film = ops.linear(temb_act, self.time_layer.weight, self.time_layer.bias)
        if film.shape[0] != h.shape[0]:
            raise NotImplementedError("time-embedding batch repeat (temporal mode) is not implemented")
        h = ops.group_norm(h, self.norm2.weight, self.norm2.bias, g, self.norm2.eps, film=film, silu=True)
        if self.config.dropout > 0 and self.training:   # reference :234 (nn.Dropout; the shipped configs use 0.0)
            h = ops.dropout(h, self.config.dropout, True)
        shortcut = x
        if self.config.output_channels != self.config.num_channels:
            shortcut = ops.conv(x, self.conv3.weight, self.conv3.bias)
        return ops.conv(h, self.conv2.weight, self.conv2.bias, residual=shortcut)


class TimeStates:
    """silu(time embedding) [B, T] plus, when they were computed up front (bf16: grouped GEMMs over all ResNets of a
    net, ops.shared_input_linears), every ResNet's FiLM projection ``time_layer(silu(temb))`` (reference :227)."""

    def __init__(self, act, film=None):
        self.act, self.film = act, film or {}


class TextStates:
    """The text conditioning as the attention layers consume it: the projected states [B, S, D] plus, when they were
    computed up front (bf16: one grouped launch for all layers, ops.text_kv), every layer's key / value projection.
    Behaves like the plain tensor for the few things the callers do with it (batch slicing in the nested model)."""

    def __init__(self, raw, kv=None):
        self.raw, self.kv = raw, kv or {}

    def __getitem__(self, idx):
        return TextStates(self.raw[idx], {k: v[idx] for k, v in self.kv.items()})

    @property
    def shape(self):
        return self.raw.shape

    def clone(self):
        return TextStates(self.raw.clone(), {k: v.clone() for k, v in self.kv.items()})

    def copy_(self, other):
        self.raw.copy_(other.raw)
        for k, v in self.kv.items():
            v.copy_(other.kv[k])
        return self


class SelfAttention(nn.Module):
    """2-D self attention + text cross attention + optional FFN (reference unet.py:241-313)."""

    def __init__(self, channels, num_heads=8, num_head_channels=-1, cond_dim=None, use_attention_ffn=False):
        super().__init__()
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0
            self.num_heads = channels // num_head_channels
        self.norm = nn.GroupNorm(32, channels)
        self.qkv = nn.Conv2d(channels, channels * 3, 1)
        self.cond_dim = cond_dim
        if cond_dim is not None and cond_dim > 0:
            self.norm_cond = nn.LayerNorm(cond_dim)
            self.kv_cond = nn.Linear(cond_dim, channels * 2)
        self.proj_out = zero_module(nn.Conv2d(channels, channels, 1))
        if use_attention_ffn:
            self.ffn = nn.Sequential(
                nn.GroupNorm(32, channels),
                nn.Conv2d(channels, 4 * channels, 1),
                nn.GELU(),
                zero_module(nn.Conv2d(4 * channels, channels, 1)),
            )
        else:
            self.ffn = None

    def forward(self, x, cond=None, cond_mask=None):
        N, H, W, C = x.shape
        hn, x = ops.group_norm(x, self.norm.weight, self.norm.bias, 32, self.norm.eps, passthrough=True)
        qkv = ops.conv(hn, self.qkv.weight, self.qkv.bias)
        kvc = None
        if self.cond_dim is not None and self.cond_dim > 0:
            if isinstance(cond, TextStates) and id(self) in cond.kv:
                kvc = cond.kv[id(self)]          # projected up front with all the other layers (ops.text_kv)
            else:
                raw = cond.raw if isinstance(cond, TextStates) else cond
                cn = ops.layer_norm(raw, self.norm_cond.weight, self.norm_cond.bias, self.norm_cond.eps)
                kvc = ops.linear(cn, self.kv_cond.weight, self.kv_cond.bias)
        a = ops.attention(qkv.reshape(N, H * W, 3 * C), kvc, cond_mask if kvc is not None else None, self.num_heads)
        a = a.reshape(N, H, W, C)
        if self.ffn is not None and ops.conv_gn_enabled() and ops.conv_gn_supported(a, self.proj_out.weight, self.ffn[0].weight, 32):
            # proj_out and the GroupNorm that opens the FFN in one launch (the norm's statistics need exactly what a
            # 256-pixel x 8-group output tile of the convolution holds)
            x, fn = ops.conv_gn(a, self.proj_out.weight, self.proj_out.bias, x, self.ffn[0].weight, self.ffn[0].bias, 32, self.ffn[0].eps)
            return ops.ffn(fn, self.ffn[1].weight, self.ffn[1].bias, self.ffn[3].weight, self.ffn[3].bias, residual=x)
        x = ops.conv(a, self.proj_out.weight, self.proj_out.bias, residual=x)
        if self.ffn is not None:
            fn, x = ops.group_norm(x, self.ffn[0].weight, self.ffn[0].bias, 32, self.ffn[0].eps, passthrough=True)
            x = ops.ffn(fn, self.ffn[1].weight, self.ffn[1].bias, self.ffn[3].weight, self.ffn[3].bias, residual=x)
        return x


class ResNetBlock(nn.Module):
    """A resolution level: ResNets (+ attention layers) and an optional down/up resample conv
    (reference unet.py:449-576)."""

    def __init__(self, temporal_dim, num_residual_blocks, num_attention_layers, downsample_output, upsample_output,
                 resnet_configs, conditioning_feature_dim=-1, temporal_mode=False, temporal_pos_emb=False,
                 temporal_spatial_ds=False, num_temporal_attention_layers=None):
        super().__init__()
        if temporal_mode or num_temporal_attention_layers:
            raise NotImplementedError("temporal (video) blocks are not implemented on the HIP path")
        assert not (downsample_output and upsample_output)
        self.num_residual_blocks = num_residual_blocks
        self.num_attention_layers = int(num_attention_layers)
        self.upsample_output, self.downsample_output = upsample_output, downsample_output
        self.level_entry = False   # set by UNet: this block is the first one at its resolution
        self.resnets = nn.ModuleList([ResNet(temporal_dim, resnet_configs[i]) for i in range(num_residual_blocks)])
        if self.num_attention_layers > 0:
            self.attn = nn.ModuleList(
                [
                    SelfAttention(resnet_configs[i].output_channels, cond_dim=conditioning_feature_dim,
                                  use_attention_ffn=resnet_configs[i].use_attention_ffn)
                    for i in range(num_residual_blocks)
                    for _ in range(self.num_attention_layers)
                ]
            )
        if downsample_output or upsample_output:
            c = resnet_configs[-1].output_channels
            self.resample = nn.Conv2d(c, c, kernel_size=3, stride=2 if downsample_output else 1, padding=1, bias=True)

    def forward(self, x, temb_act, skip_activations=None, return_activations=False, conditioning=None, cond_mask=None,
                tap=None):
        """``tap``: where the incoming x is recorded as a skip activation, if it is (see ResNet.forward)"""
        if self.level_entry:
            # backward reaches this point once every layer of this and the previous resolution level is done: queued
            # same-shape weight gradients of that level go out as grouped launches here (ops.flush_wgrad_queue)
            if tap is not None and tap[0][tap[1]] is x:
                x = tap[0][tap[1]] = ops.wgrad_flush_point(x)
            else:
                x = ops.wgrad_flush_point(x)
        activations = []
        L = self.num_attention_layers
        for i in range(self.num_residual_blocks):
            if skip_activations is not None:
                x = ops.concat(x, skip_activations.pop(0))
                tap = None
            x = self.resnets[i](x, temb_act, tap)
            tap = None
            for j in range(L):
                x = self.attn[i * L + j](x, conditioning, cond_mask)
            activations.append(x)
            if skip_activations is None:
                tap = (activations, len(activations) - 1)   # the next ResNet of this block reads the x recorded here
        if self.downsample_output:
            if activations and activations[-1] is x:
                # x also feeds a skip connection (reference unet.py:566-567): that gradient joins the convolution's own
                x, activations[-1] = ops.conv_tap(x, self.resample.weight, self.resample.bias, stride=2)
            else:
                x = ops.conv(x, self.resample.weight, self.resample.bias, stride=2)
            activations.append(x)
        elif self.upsample_output:
            x = ops.upsample_conv(x, self.resample.weight, self.resample.bias)
            activations.append(x)
        return (x, activations) if return_activations else x


# --------------------------------------------------------------------------------------
# LM head: self-attention blocks over the text states (reference unet.py:316-387, 425-446; ``num_lm_head_layers`` > 0).
# Off the denoiser's hot path -- B x 32 tokens, once per sampling run / train step -- and no shipped config enables it:
# LayerNorm and the linear layers are this package's kernels, the S x S token attention and the GELU are torch ops on the
# same GPU tensors ("reference-PyTorch fallback" for these blocks).  Rotary position embeddings
# (``pos_emb``, a third-party package in the reference) are only used by the temporal blocks and are not implemented.
# --------------------------------------------------------------------------------------
class SelfAttention1D(nn.Module):
    def __init__(self, channels, num_heads=8, num_head_channels=-1, use_attention_ffn=False, pos_emb=False):
        super().__init__()
        if pos_emb:
            raise NotImplementedError("rotary position embeddings (temporal blocks) are not implemented on the HIP path")
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0
            self.num_heads = channels // num_head_channels
        self.norm = nn.LayerNorm(channels)
        self.qkv = nn.Linear(channels, channels * 3)
        self.proj_out = zero_module(nn.Linear(channels, channels))
        self.ffn = None
        if use_attention_ffn:
            self.ffn = nn.Sequential(nn.LayerNorm(channels), nn.Linear(channels, 4 * channels), nn.GELU(),
                                     zero_module(nn.Linear(4 * channels, channels)))

    def attention(self, q, k, v, mask=None):
        bs, length, width = q.shape
        ch = width // self.num_heads
        scale = 1 / math.sqrt(math.sqrt(ch))
        q = q.reshape(bs, length, self.num_heads, ch)
        k = k.reshape(bs, length, self.num_heads, ch)
        weight = torch.einsum("bthc,bshc->bhts", q * scale, k * scale)
        if mask is not None:
            weight = weight.masked_fill(mask.view(mask.size(0), 1, 1, mask.size(1)) == 0, float("-inf"))
        weight = torch.softmax(weight.float(), dim=-1).type(weight.dtype)
        a = torch.einsum("bhts,bshc->bthc", weight, v.reshape(bs, -1, self.num_heads, ch))
        return a.reshape(bs, length, -1)

    def forward(self, x, mask):
        qkv = ops.linear(ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps), self.qkv.weight, self.qkv.bias)
        q, k, v = qkv.chunk(3, dim=-1)
        h = self.attention(q, k, v, mask).contiguous()
        x = ops.linear(h, self.proj_out.weight, self.proj_out.bias, residual=x)
        if self.ffn is not None:
            n, l1, _, l2 = self.ffn
            h = F.gelu(ops.linear(ops.layer_norm(x, n.weight, n.bias, n.eps), l1.weight, l1.bias))
            x = ops.linear(h, l2.weight, l2.bias, residual=x)
        return x


class MLP(nn.Module):
    def __init__(self, channels, multiplier=4):
        super().__init__()
        self.main = nn.Sequential(nn.LayerNorm(channels), nn.Linear(channels, multiplier * channels), nn.GELU(),
                                  zero_module(nn.Linear(multiplier * channels, channels)))

    def forward(self, x):
        n, l1, _, l2 = self.main
        h = F.gelu(ops.linear(ops.layer_norm(x, n.weight, n.bias, n.eps), l1.weight, l1.bias))
        return ops.linear(h, l2.weight, l2.bias, residual=x)


class SelfAttention1DBlock(nn.Module):
    def __init__(self, channels, num_heads=8, num_head_channels=-1, mlp_multiplier=4):
        super().__init__()
        self.attn = SelfAttention1D(channels, num_heads, num_head_channels)
        self.mlp = MLP(channels, mlp_multiplier)

    def forward(self, x, mask):
        return self.mlp(self.attn(x, mask))


# --------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------
class UNet(nn.Module):
    def __init__(self, input_channels: int, output_channels: int, config: UNetConfig):
        super().__init__()
        if config.temporal_mode:
            raise NotImplementedError("temporal_mode is not implemented on the HIP path")
        self.config = config
        self.input_channels, self.output_channels = input_channels, output_channels
        self.input_conditioning_feature_dim = config.conditioning_feature_dim
        if config.conditioning_feature_dim > 0 and config.conditioning_feature_proj_dim > 0:
            config.conditioning_feature_dim = config.conditioning_feature_proj_dim  # same side effect as the reference (:589-593)
        chans = config.resolution_channels
        self.temporal_dim = chans[0] * 4 if config.temporal_dim is None else config.temporal_dim

        half = self.temporal_dim // 8
        freqs = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / half))
        self.register_buffer("t_emb", freqs.unsqueeze(0), persistent=False)
        self.temb_layer1 = nn.Linear(self.temporal_dim // 4, self.temporal_dim)
        self.temb_layer2 = nn.Linear(self.temporal_dim, self.temporal_dim)

        has_cond = config.conditioning_feature_dim > 0 and not config.skip_cond_emb
        self.cond_emb = nn.Linear(config.conditioning_feature_dim, self.temporal_dim, bias=False) if has_cond else None

        self.conditions = None
        if config.micro_conditioning is not None:
            self.conditions = {c.split(":")[0]: float(c.split(":")[1]) for c in config.micro_conditioning.split(",")}
            self.cond_layers = nn.ModuleDict(
                {
                    name: nn.ModuleList(
                        [
                            nn.Linear(self.temporal_dim // 4, self.temporal_dim),
                            zero_module(nn.Linear(self.temporal_dim, self.temporal_dim)),
                        ]
                    )
                    for name in self.conditions
                }
            )

        c = chans[0]
        self.conv_in = nn.Conv2d(input_channels, c, kernel_size=3, stride=1, padding=1, bias=True)
        skip_channels = [c]
        nres = len(chans)
        self.num_resolutions = nres

        def level_cfg(cin, cout):
            rc = copy.copy(config.resnet_config)
            rc.num_channels, rc.output_channels = cin, cout
            return rc

        def n_attn(i):
            return config.num_attention_layers[i] if i in config.attention_levels else 0

        def cdim(i):
            return config.conditioning_feature_dim if i in config.attention_levels else -1

        down = []
        for i in range(nres):
            cfgs = []
            for _ in range(config.num_resnets_per_resolution[i]):
                cfgs.append(level_cfg(c, chans[i]))
                c = chans[i]
                skip_channels.append(c)
            if i != nres - 1:
                skip_channels.append(c)
            down.append(
                ResNetBlock(self.temporal_dim, len(cfgs), n_attn(i), downsample_output=i != nres - 1,
                            upsample_output=False, resnet_configs=cfgs, conditioning_feature_dim=cdim(i))
            )
        if not config.skip_mid_blocks:
            mid = [
                ResNetBlock(self.temporal_dim, 1, 1, False, False, resnet_configs=[level_cfg(c, c)],
                            conditioning_feature_dim=config.conditioning_feature_dim),
                ResNetBlock(self.temporal_dim, 1, 0, False, False, resnet_configs=[level_cfg(c, c)]),
            ]
        up = []
        for i in reversed(range(nres)):
            cfgs = []
            for _ in range(config.num_resnets_per_resolution[i] + 1):
                cfgs.append(level_cfg(c + skip_channels.pop(), chans[i]))
                c = chans[i]
            up.append(
                ResNetBlock(self.temporal_dim, len(cfgs), n_attn(i), downsample_output=False, upsample_output=i != 0,
                            resnet_configs=cfgs, conditioning_feature_dim=cdim(i))
            )
        self.norm_out = nn.GroupNorm(config.resnet_config.num_groups_norm, c)
        self.conv_out = zero_module(nn.Conv2d(c, output_channels, kernel_size=3, padding=1))
        self._config = config
        self.down_blocks = nn.ModuleList(down)
        if not config.skip_mid_blocks:
            self.mid_blocks = nn.ModuleList(mid)
        self.up_blocks = nn.ModuleList(up)

        for i in range(1, nres):   # blocks that open a new resolution level (after a down- / up-sampling conv)
            self.down_blocks[i].level_entry = True
            self.up_blocks[i].level_entry = True
        self.masked_cross_attention = config.masked_cross_attention
        if has_cond:
            if config.conditioning_feature_proj_dim > 0:
                self.lm_proj = nn.Linear(self.input_conditioning_feature_dim, config.conditioning_feature_dim)
            self.lm_head = nn.ModuleList([SelfAttention1DBlock(config.conditioning_feature_dim)
                                          for _ in range(config.num_lm_head_layers)])
        self.is_temporal = []

    # ---- bookkeeping identical in behaviour to the reference ---------------------------
    @property
    def model_type(self):
        return "unet"

    def print_size(self, target_image_size: int = 64):
        n = sum(p.numel() for p in self.parameters())
        logging.info("%s: %.1f M parameters", type(self).__name__, n / 1e6)

    def save(self, fname: str, other_items=None):
        """Same container as the reference (unet.py:794-800): {"state_dict": ..., **other_items}."""
        ckpt = {"state_dict": self.state_dict()}
        ckpt.update(other_items or {})
        torch.save(ckpt, fname)

    def load(self, fname: str):
        """Non-strict load on the key intersection; returns the non-weight items (unet.py:802-832)."""
        # tensors-only unpickling first; the reference's checkpoints may carry plain-Python extras that need the full
        # unpickler (reference unet.py:803 uses it unconditionally) -- which executes code from the file: trusted files only
        try:
            ckpt = torch.load(fname, map_location="cpu", weights_only=True)
        except Exception:
            if os.environ.get("MDM_HIP_SAFE_LOAD", "0") == "1":
                raise
            ckpt = torch.load(fname, map_location="cpu", weights_only=False)
        mine = self.state_dict()
        theirs = ckpt["state_dict"]
        common = {k: v for k, v in theirs.items() if k in mine}
        extra, missing = set(theirs) - set(mine), set(mine) - set(common)
        if extra or missing:
            print(extra, missing)
        self.load_state_dict(common, strict=False)
        return {k: copy.copy(v) for k, v in ckpt.items() if k != "model_state_dict"}

    # ---- forward pieces --------------------------------------------------------------------
    def create_temporal_embedding(self, times, ff_layers=None):
        dt = compute_dtype()
        emb = ops.sincos_embedding(times.reshape(-1).float(), self.t_emb, dt)
        l1, l2 = (self.temb_layer1, self.temb_layer2) if ff_layers is None else ff_layers
        h = ops.silu(ops.linear(emb, l1.weight, l1.bias))
        return ops.linear(h, l2.weight, l2.bias)

    def forward_conditioning(self, conditioning, cond_mask):
        cond = ops.cast(conditioning, compute_dtype())
        if self.config.conditioning_feature_proj_dim > 0:
            cond = ops.linear(cond, self.lm_proj.weight, self.lm_proj.bias)
        for head in self.lm_head:   # reference :850-853
            cond = head(cond, cond_mask if self.masked_cross_attention else None)
        # reference :854-861: with an LM head and unmasked cross attention the mean runs over ALL tokens
        y = ops.masked_mean(cond, None if (not self.masked_cross_attention and len(self.lm_head) > 0) else cond_mask)
        if not self.masked_cross_attention:
            cond_mask = None
        cond_emb = ops.linear(y, self.cond_emb.weight, None)
        return cond_emb, self.text_states(cond), cond_mask

    def text_states(self, cond):
        """every attention layer applies its own LayerNorm + Linear to the SAME text states (reference :263-264, 304):
        in bf16 all of them are computed here, in one multi-norm launch and one grouped GEMM per width; sampling loops
        that call forward_conditioning once (GraphedSampler) thereby hoist them out of the denoising loop"""
        layers = [m for m in self.modules() if isinstance(m, SelfAttention) and m.cond_dim is not None and m.cond_dim > 0]
        quads = [(m.norm_cond.weight, m.norm_cond.bias, m.kv_cond.weight, m.kv_cond.bias) for m in layers]
        if not layers or not ops.text_kv_supported(cond, quads) or len({m.norm_cond.eps for m in layers}) != 1:
            return TextStates(cond)
        kv = ops.text_kv(cond, quads, layers[0].norm_cond.eps)
        return TextStates(cond, {id(m): t for m, t in zip(layers, kv)})

    def late_gradient_parameters(self):
        """parameters whose gradients are complete only when backward ENDS although they are registered inside the
        blocks: the layers evaluated for the whole network in one grouped launch at the start of forward (time_states,
        text_states).  A gradient reducer puts them at the end of its arena (mdm_hip.distributed.GradReducer)."""
        out = []
        for m in self.modules():
            if isinstance(m, ResNet):
                out += list(m.time_layer.parameters())
            elif isinstance(m, SelfAttention) and m.cond_dim is not None and m.cond_dim > 0:
                out += list(m.norm_cond.parameters()) + list(m.kv_cond.parameters())
        return out

    def forward_micro_conditioning(self, times, micros):
        temb = None
        for key, default in self.conditions.items():
            micro = micros.get(key, default * torch.ones_like(times))
            micro = (micro / default).clamp(max=1) * default if key == "scale" else micro * 1000
            t = self.create_temporal_embedding(micro, ff_layers=self.cond_layers[key])
            temb = t if temb is None else ops.add(temb, t)
        return temb

    def time_states(self, temb):
        """silu(temb) and the FiLM projections of every ResNet of THIS net (not of a nested inner one), all at once"""
        act = ops.silu(temb)
        blocks = list(self.down_blocks) + (list(self.mid_blocks) if hasattr(self, "mid_blocks") else []) + list(self.up_blocks)
        resnets = [r for b in blocks for r in b.resnets]
        pairs = [(r.time_layer.weight, r.time_layer.bias) for r in resnets]
        if not ops.shared_input_linears_supported(act, pairs):
            return TimeStates(act)
        films = ops.shared_input_linears(act, pairs)
        return TimeStates(act, {id(r): f for r, f in zip(resnets, films)})

    def _time_embedding(self, times, cond_emb, micros):
        temb = self.create_temporal_embedding(times)
        if cond_emb is not None:
            temb = ops.add(temb, cond_emb)
        if self.conditions is not None:
            temb = ops.add(temb, self.forward_micro_conditioning(times, micros))
        return temb

    def forward_input_layer(self, x_t, normalize=False):
        if isinstance(x_t, (list, tuple)) and len(x_t) == 1:
            x_t = x_t[0]
        if normalize:   # per-sample x / std (reference :871-872), its own kernel pair
            x_t = ops.sample_std_normalize(x_t.float())
        return ops.conv(ops.to_nhwc(x_t, compute_dtype()), self.conv_in.weight, self.conv_in.bias)

    def forward_output_layer(self, x):
        h = ops.group_norm(x, self.norm_out.weight, self.norm_out.bias, self.norm_out.num_groups, self.norm_out.eps, silu=True)
        return ops.from_nhwc(ops.conv(h, self.conv_out.weight, self.conv_out.bias), self.output_channels)

    def forward_downsample(self, x, temb_act, conditioning, cond_mask):
        skips = [x]
        for i, block in enumerate(self.down_blocks):
            tap = (skips, len(skips) - 1)   # the x this block starts from is the last recorded skip activation
            if i in self.config.attention_levels:
                x, acts = block(x, temb_act, return_activations=True, conditioning=conditioning, cond_mask=cond_mask, tap=tap)
            else:
                x, acts = block(x, temb_act, return_activations=True, tap=tap)
            skips.extend(acts)
        return x, skips

    def forward_upsample(self, x, temb_act, conditioning, cond_mask, skips):
        nres = len(self._config.resolution_channels)
        for i, block in enumerate(self.up_blocks):
            ri = nres - 1 - i
            k = self._config.num_resnets_per_resolution[ri] + 1
            mine = skips[-k:][::-1]
            del skips[-k:]
            if ri in self.config.attention_levels:
                x = block(x, temb_act, skip_activations=mine, conditioning=conditioning, cond_mask=cond_mask)
            else:
                x = block(x, temb_act, skip_activations=mine)
        return x

    def forward_denoising(self, x_t, times, cond_emb=None, conditioning=None, cond_mask=None, micros={}):
        """When ``config.nesting`` the input is ``(x_t, x_feat)`` with ``x_feat`` an NHWC feature map
        from the enclosing NestedUNet, and the result is ``(x_out, features)`` (reference :946-968)."""
        temb_act = self.time_states(self._time_embedding(times, cond_emb, micros))
        if self._config.nesting:
            x_t, x_feat = x_t
        x = self.forward_input_layer(x_t)
        if self._config.nesting and x_feat is not None:
            x = ops.add(x, x_feat)
        x, skips = self.forward_downsample(x, temb_act, conditioning, cond_mask)
        if not self.config.skip_mid_blocks:
            x = self.mid_blocks[0](x, temb_act, conditioning=conditioning, cond_mask=cond_mask, tap=(skips, len(skips) - 1))
            x = self.mid_blocks[1](x, temb_act)
        x = self.forward_upsample(x, temb_act, conditioning, cond_mask, skips)
        x_out = self.forward_output_layer(x)
        return (x_out, x) if self._config.nesting else x_out

    def forward(self, x_t, times, conditioning=None, cond_mask=None, micros={}):
        if self.config.conditioning_feature_dim > 0:
            cond_emb, conditioning, cond_mask = self.forward_conditioning(conditioning, cond_mask)
        else:
            cond_emb = None
        return self.forward_denoising(x_t, times, cond_emb, conditioning, cond_mask, micros)

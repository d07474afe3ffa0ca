"""CPU oracle for the (Nested)U-Net denoiser hot path -- TEST INFRASTRUCTURE ONLY.

A functional, state_dict-driven restatement (plain torch ops on CPU, fp32 or fp64,
NCHW like the reference) of what apple/ml-mdm's ``UNet`` / ``NestedUNet`` compute.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file; the product package never does.

Pinning: the reference holds no golden vectors for this path, so
the oracle is pinned against outputs of the reference itself, generated in the build
container by ``oracle/make_golden.py`` (which imports /root/reference through
``oracle/ref_import.py``) and committed under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks oracle == reference on those vectors.

Every function cites the reference lines it restates (paths relative to
ml-mdm-matryoshka/ml_mdm/).
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------
# small pieces
# ---------------------------------------------------------------------------------------
def _conv(sd, name, x, stride=1):
    w = sd[name + ".weight"]
    pad = (w.shape[-1] - 1) // 2
    return F.conv2d(x, w, sd.get(name + ".bias"), stride=stride, padding=pad)


def _linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _gn(sd, name, x, groups, eps=1e-5):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)


def timestep_features(times, temporal_dim):
    """models/unet.py:600-602, 835-836: [sin(t*f) | cos(t*f)], f_i = 10000^(-i/half), half = temporal_dim/8."""
    half = temporal_dim // 8
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / half)).to(times.device)
    ang = times.reshape(-1, 1).to(torch.float32) * freqs.reshape(1, -1)
    return torch.cat([ang.sin(), ang.cos()], dim=1)


def time_mlp(sd, l1, l2, feats):
    """models/unet.py:840-845: Linear -> SiLU -> Linear."""
    feats = feats.to(sd[l1 + ".weight"].dtype)
    return _linear(sd, l2, F.silu(_linear(sd, l1, feats)))


def attention_core(q, k, v, heads, mask=None):
    """models/unet.py:276-294.  q: [B, C, L], k/v: [B, C, S]; head h owns channels [h*d, (h+1)*d);
    both q and k are scaled by d^-1/4; softmax over keys in fp32 (or wider); masked keys -> -inf."""
    B, C, L = q.shape
    d = C // heads
    s = 1.0 / math.sqrt(math.sqrt(d))
    qh = (q * s).reshape(B * heads, d, L)
    kh = (k * s).reshape(B * heads, d, -1)
    vh = v.reshape(B * heads, d, -1)
    logits = torch.bmm(qh.transpose(1, 2), kh)  # [BH, L, S]
    if mask is not None:
        dead = (mask == 0).reshape(B, 1, 1, -1).expand(B, heads, 1, mask.shape[1]).reshape(B * heads, 1, -1)
        logits = logits.masked_fill(dead, float("-inf"))
    acc_dtype = torch.float64 if logits.dtype == torch.float64 else torch.float32
    probs = torch.softmax(logits.to(acc_dtype), dim=-1).to(logits.dtype)
    out = torch.bmm(vh, probs.transpose(1, 2))  # [BH, d, L]
    return out.reshape(B, C, L)


def attention_layer(sd, pfx, x, cond, cond_mask, heads=8):
    """models/unet.py:296-313: GN -> qkv 1x1 -> self-attn (+ separate cross-attn over LayerNorm'd text,
    summed) -> zero-init proj + residual -> optional GN / 1x1 / GELU(erf) / 1x1 FFN + residual."""
    B, C, H, W = x.shape
    qkv = _conv(sd, pfx + "qkv", _gn(sd, pfx + "norm", x, 32)).reshape(B, 3 * C, H * W)
    q, k, v = qkv[:, :C], qkv[:, C : 2 * C], qkv[:, 2 * C :]
    h = attention_core(q, k, v, heads)
    if (pfx + "kv_cond.weight") in sd:
        D = cond.shape[-1]
        cn = F.layer_norm(cond, (D,), sd[pfx + "norm_cond.weight"], sd[pfx + "norm_cond.bias"], 1e-5)
        kv = _linear(sd, pfx + "kv_cond", cn).transpose(1, 2)  # [B, 2C, S]
        h = h + attention_core(q, kv[:, :C], kv[:, C:], heads, cond_mask)
    x = x + _conv(sd, pfx + "proj_out", h.reshape(B, C, H, W))
    if (pfx + "ffn.1.weight") in sd:
        f = _conv(sd, pfx + "ffn.1", _gn(sd, pfx + "ffn.0", x, 32))
        x = x + _conv(sd, pfx + "ffn.3", F.gelu(f))
    return x


def resnet(sd, pfx, x, temb, groups):
    """models/unet.py:223-238."""
    h = _conv(sd, pfx + "conv1", F.silu(_gn(sd, pfx + "norm1", x, groups)))
    film = _linear(sd, pfx + "time_layer", F.silu(temb))
    cout = film.shape[1] // 2
    ta, tb = film[:, :cout, None, None], film[:, cout:, None, None]
    h = F.silu(_gn(sd, pfx + "norm2", h, groups) * (1 + ta) + tb)
    h = _conv(sd, pfx + "conv2", h)
    if (pfx + "conv3.weight") in sd:
        x = _conv(sd, pfx + "conv3", x)
    return x + h


def level(sd, pfx, x, temb, groups, n_res, n_attn, mode, cond, cond_mask, skips=None):
    """One ResNetBlock (models/unet.py:534-576).  mode in {"down", "up", "none"}; returns (x, activations)."""
    acts = []
    for i in range(n_res):
        if skips is not None:
            x = torch.cat([x, skips.pop(0)], dim=1)
        x = resnet(sd, "%sresnets.%d." % (pfx, i), x, temb, groups)
        for j in range(n_attn):
            x = attention_layer(sd, "%sattn.%d." % (pfx, i * n_attn + j), x, cond, cond_mask)
        acts.append(x)
    if mode == "down":
        x = _conv(sd, pfx + "resample", x, stride=2)
        acts.append(x)
    elif mode == "up":
        x = _conv(sd, pfx + "resample", F.interpolate(x, scale_factor=2, mode="nearest"))
        acts.append(x)
    return x, acts


# ---------------------------------------------------------------------------------------
# whole models
# ---------------------------------------------------------------------------------------
def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def _temporal_dim(cfg):
    return cfg.resolution_channels[0] * 4 if cfg.temporal_dim is None else cfg.temporal_dim


def _micro_conditions(cfg):
    if cfg.micro_conditioning is None:
        return None
    return {c.split(":")[0]: float(c.split(":")[1]) for c in cfg.micro_conditioning.split(",")}


def time_embedding(sd, cfg, times, cond_emb, micros):
    """models/unet.py:938-943 + 920-933."""
    tdim = _temporal_dim(cfg)
    temb = time_mlp(sd, "temb_layer1", "temb_layer2", timestep_features(times, tdim))
    if cond_emb is not None:
        temb = temb + cond_emb
    conds = _micro_conditions(cfg)
    if conds is not None:
        for key, default in conds.items():
            m = micros.get(key, default * torch.ones_like(times))
            m = (m / default).clamp(max=1) * default if key == "scale" else m * 1000
            temb = temb + time_mlp(sd, "cond_layers.%s.0" % key, "cond_layers.%s.1" % key, timestep_features(m, tdim))
    return temb


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def lm_head_block(sd, prefix, x, mask, heads=8):
    """One SelfAttention1DBlock over the text tokens x [B, S, C] (models/unet.py:439-446): SelfAttention1D without rotary
    embedding and without its own ffn (:316-387: LayerNorm, qkv, 8 heads of C / 8 channels, q and k scaled by d^-1/4,
    key mask -> -inf, softmax in fp32 or wider, proj_out, residual), then MLP (:425-436: x + Linear(GELU(Linear(LN(x)))))."""
    B, S, C = x.shape
    d = C // heads
    q, k, v = _linear(sd, prefix + ".attn.qkv", _ln(sd, prefix + ".attn.norm", x)).chunk(3, dim=-1)
    sc = 1.0 / math.sqrt(math.sqrt(d))
    qh, kh, vh = (t.reshape(B, S, heads, d).permute(0, 2, 1, 3) for t in (q * sc, k * sc, v))   # [B, heads, S, d]
    w = qh @ kh.transpose(-1, -2)                                                             # [B, heads, query, key]
    if mask is not None:
        w = w.masked_fill(mask.reshape(B, 1, 1, S) == 0, float("-inf"))
    w = torch.softmax(w.to(torch.promote_types(w.dtype, torch.float32)), dim=-1).to(w.dtype)
    a = (w @ vh).permute(0, 2, 1, 3).reshape(B, S, C)
    x = x + _linear(sd, prefix + ".attn.proj_out", a)
    m = prefix + ".mlp.main"
    return x + _linear(sd, m + ".3", F.gelu(_linear(sd, m + ".1", _ln(sd, m + ".0", x))))


def conditioning_path(sd, cfg, conditioning, cond_mask):
    """models/unet.py:847-865."""
    if (cfg.conditioning_feature_proj_dim or -1) > 0:
        conditioning = _linear(sd, "lm_proj", conditioning)
    n_head = int(getattr(cfg, "num_lm_head_layers", 0) or 0)
    for i in range(n_head):
        conditioning = lm_head_block(sd, "lm_head.%d" % i, conditioning, cond_mask if cfg.masked_cross_attention else None)
    if cond_mask is None or (not cfg.masked_cross_attention and n_head > 0):
        y = conditioning.mean(dim=1)
    else:
        y = (cond_mask.unsqueeze(-1) * conditioning).sum(dim=1) / cond_mask.sum(dim=1, keepdim=True)
    if not cfg.masked_cross_attention:
        cond_mask = None
    return F.linear(y, sd["cond_emb.weight"]), conditioning, cond_mask


def _down(sd, cfg, x, temb, cond, cond_mask):
    """models/unet.py:883-897."""
    groups = cfg.resnet_config.num_groups_norm
    nres = len(cfg.resolution_channels)
    skips = [x]
    for i in range(nres):
        att = cfg.num_attention_layers[i] if i in cfg.attention_levels else 0
        x, acts = level(sd, "down_blocks.%d." % i, x, temb, groups, cfg.num_resnets_per_resolution[i], att,
                        "down" if i != nres - 1 else "none", cond, cond_mask)
        skips.extend(acts)
    return x, skips


def _up(sd, cfg, x, temb, cond, cond_mask, skips):
    """models/unet.py:900-918."""
    groups = cfg.resnet_config.num_groups_norm
    nres = len(cfg.resolution_channels)
    for bi, i in enumerate(reversed(range(nres))):
        k = cfg.num_resnets_per_resolution[i] + 1
        mine = skips[-k:][::-1]
        del skips[-k:]
        att = cfg.num_attention_layers[i] if i in cfg.attention_levels else 0
        x, _ = level(sd, "up_blocks.%d." % bi, x, temb, groups, k, att, "up" if i != 0 else "none", cond, cond_mask, skips=mine)
    return x


def _head(sd, cfg, x):
    """models/unet.py:877-880."""
    return _conv(sd, "conv_out", F.silu(_gn(sd, "norm_out", x, cfg.resnet_config.num_groups_norm)))


def unet_denoise(sd, cfg, x_t, times, cond_emb, conditioning, cond_mask, micros):
    """models/unet.py:935-969."""
    temb = time_embedding(sd, cfg, times, cond_emb, micros)
    x_feat = None
    if cfg.nesting:
        x_t, x_feat = x_t
    if isinstance(x_t, (list, tuple)) and len(x_t) == 1:
        x_t = x_t[0]
    x = _conv(sd, "conv_in", x_t)
    if x_feat is not None:
        x = x + x_feat
    x, skips = _down(sd, cfg, x, temb, conditioning, cond_mask)
    if not cfg.skip_mid_blocks:
        groups = cfg.resnet_config.num_groups_norm
        x, _ = level(sd, "mid_blocks.0.", x, temb, groups, 1, 1, "none", conditioning, cond_mask)
        x, _ = level(sd, "mid_blocks.1.", x, temb, groups, 1, 0, "none", None, None)
    x = _up(sd, cfg, x, temb, conditioning, cond_mask, skips)
    out = _head(sd, cfg, x)
    return (out, x) if cfg.nesting else out


def nested_denoise(sd, cfg, x_t, times, cond_emb, conditioning, cond_mask, micros):
    """models/nested_unet.py:168-230 (equal batch sizes at every scale, or bh < bl zero padding)."""
    temb = time_embedding(sd, cfg, times, cond_emb, micros)
    x_feat = None
    if cfg.nesting:
        x_t, x_feat = x_t
    bh, bl = x_t[0].shape[0], x_t[1].shape[0]
    x_low_in, x_hi = x_t[1:], x_t[0]
    if not cfg.skip_normalization:
        x_hi = x_hi / x_hi.std((1, 2, 3), keepdim=True)
    x = _conv(sd, "conv_in", x_hi)
    if x_feat is not None:
        x = x + x_feat
    cond_hi = conditioning[:bh]
    mask_hi = cond_mask[:bh] if cond_mask is not None else None
    x, skips = _down(sd, cfg, x, temb[:bh], cond_hi, mask_hi)
    x_inner = _conv(sd, "in_adapter", x) if "in_adapter.weight" in sd else None
    if x_inner is not None and bh < bl:
        x_inner = torch.cat([x_inner, x_inner.new_zeros(bl - bh, *x_inner.shape[1:])], 0)
    inner_sd, inner_cfg = _sub(sd, "inner_unet."), cfg.inner_config
    inner_fn = nested_denoise if getattr(inner_cfg, "inner_config", None) is not None else unet_denoise
    x_low, x_inner = inner_fn(inner_sd, inner_cfg, (x_low_in, x_inner), times, cond_emb, conditioning, cond_mask, micros)
    x_inner = _conv(sd, "out_adapter", x_inner)
    x = x + (x_inner[:bh] if bh < bl else x_inner)
    x = _up(sd, cfg, x, temb[:bh], cond_hi, mask_hi, skips)
    out = _head(sd, cfg, x)
    outs = [out] + x_low if isinstance(x_low, list) else [out, x_low]
    return (outs, x) if cfg.nesting else outs


def model_forward(sd, cfg, x_t, times, conditioning=None, cond_mask=None, micros=None):
    """models/unet.py:971-987 / nested_unet.py:165-166: top-level forward of either model type."""
    micros = micros or {}
    nested = getattr(cfg, "inner_config", None) is not None
    # the conditioning path lives in the innermost UNet
    c_sd, c_cfg = sd, cfg
    while getattr(c_cfg, "inner_config", None) is not None:
        c_sd, c_cfg = _sub(c_sd, "inner_unet."), c_cfg.inner_config
    cond_emb = None
    if "cond_emb.weight" in c_sd:
        cond_emb, conditioning, cond_mask = conditioning_path(c_sd, c_cfg, conditioning, cond_mask)
    fn = nested_denoise if nested else unet_denoise
    return fn(sd, cfg, x_t, times, cond_emb, conditioning, cond_mask, micros)


# ---------------------------------------------------------------------------------------
# helpers shared by the parity tests
# ---------------------------------------------------------------------------------------
def randomize_zero_params(state_dict, seed=4321):
    """Fresh reference init zero-fills conv2 / proj_out / ffn.3 / conv_out / adapters / cond_layers.*.1 and the
    norm biases, which makes outputs and 99.7% of gradients vanish.  Replace every
    all-zero tensor by seeded noise: sigma = 1/sqrt(fan_in) for weights, 0.02 for vectors."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(state_dict):
        v = state_dict[k]
        if v.dtype.is_floating_point and v.numel() > 0 and float(v.abs().max()) == 0.0:
            if v.dim() >= 2:
                fan_in = v[0].numel()
                v = torch.randn(v.shape, generator=g, dtype=torch.float32) / math.sqrt(fan_in)
            else:
                v = 0.02 * torch.randn(v.shape, generator=g, dtype=torch.float32)
        out[k] = v.clone()
    return out


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

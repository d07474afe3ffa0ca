"""Seeded parity cases shared by the oracle/golden tests (CPU) and the HIP parity tests (GPU).

A case is fully determined by its name: architecture config, seeded parameter values
(reference init + seeded randomisation of the zero-initialised tensors), seeded inputs.
torch's CPU generator is bit-reproducible for a fixed torch build, and both boxes run the
same image, so the golden files only need to carry *outputs* -- plus per-tensor checksums
of the parameters to prove the regeneration matched.
"""
import torch

import unet_oracle as O


def _cfg(name):
    from mdm_hip import configs

    return {
        "mini_unet": configs.mini_unet_config,
        "mini_unet_masked": lambda: configs.mini_unet_config(masked=1),
        "mini_nested": configs.mini_nested_config,
    }[name]()


CASES = ["mini_unet", "mini_unet_masked", "mini_nested"]


def build_module(name, seed=0):
    """our (product) module, constructed on CPU, with the case's parameter values loaded"""
    import mdm_hip

    cfg = _cfg(name)
    cls = mdm_hip.NestedUNet if hasattr(cfg, "inner_config") else mdm_hip.UNet
    torch.manual_seed(seed)
    model = cls(3, 3, cfg)
    sd = O.randomize_zero_params(model.state_dict(), seed=4321 + seed)
    model.load_state_dict(sd)
    return model, _cfg(name), sd


def inputs(name, seed=1):
    g = torch.Generator().manual_seed(seed)
    B, S, D = 2, 8, 64
    cond = torch.randn(B, S, D, generator=g)
    mask = torch.ones(B, S)
    mask[1, 5:] = 0
    cond = cond * mask.unsqueeze(-1)
    times = torch.tensor([3, 977])
    side = 32 if name == "mini_nested" else 16
    if name == "mini_nested":
        x = [torch.randn(B, 3, side, side, generator=g), torch.randn(B, 3, side // 2, side // 2, generator=g)]
    else:
        x = torch.randn(B, 3, side, side, generator=g)
    outs = [(B, 3, side, side)] + ([(B, 3, side // 2, side // 2)] if name == "mini_nested" else [])
    gys = [torch.randn(s, generator=g) for s in outs]
    return dict(x=x, times=times, cond=cond, mask=mask, gys=gys)


def probe_for(key, shape):
    """deterministic probe tensor used to summarise a gradient as one number"""
    g = torch.Generator().manual_seed(abs(hash_str(key)) % (2**31))
    return torch.randn(shape, generator=g, dtype=torch.float64)


def hash_str(s):
    h = 1469598103934665603
    for ch in s.encode():
        h = ((h ^ ch) * 1099511628211) % (2**64)
    return h


def as_list(y):
    return list(y) if isinstance(y, (list, tuple)) else [y]


def loss_of(outs, gys):
    return sum((o.double() * g.double().to(o.device)).sum() for o, g in zip(as_list(outs), gys))


def oracle_run(name, dtype=torch.float32, with_grad=True):
    """oracle forward (+ parameter gradients) on CPU"""
    _, cfg, sd = build_module(name)
    inp = inputs(name)
    leaf = {k: v.to(dtype).clone().requires_grad_(with_grad) for k, v in sd.items()}
    cast = lambda t: [u.to(dtype) for u in t] if isinstance(t, list) else t.to(dtype)
    outs = O.model_forward(leaf, cfg, cast(inp["x"]), inp["times"], inp["cond"].to(dtype), inp["mask"].to(dtype))
    grads = None
    if with_grad:
        loss_of(outs, inp["gys"]).backward()
        grads = {k: v.grad for k, v in leaf.items()}
    return [o.detach() for o in as_list(outs)], grads


def grad_errors(grads, g_ref, floor_frac=1e-2):
    """per-parameter ||g - g_ref|| / max(||g_ref||, floor).  Gradients that are mathematically zero (e.g. the
    bias of a conv feeding a 1-channel-per-group GroupNorm) are rounding noise on both sides, so every tensor
    is measured against a floor of ``floor_frac`` x the median gradient norm rather than its own noise norm."""
    norms = sorted(float(g.double().norm()) for g in g_ref.values())
    floor = floor_frac * norms[len(norms) // 2]
    errs = {}
    for k, r in g_ref.items():
        d = float((grads[k].detach().double().cpu() - r.double()).norm())
        errs[k] = d / max(float(r.double().norm()), floor)
    return errs, floor

"""Helpers for benchmarks and tools that must not import the oracle: seeded parameter randomisation."""
import math

import torch


def randomize_zero_params(state_dict, seed=4321):
    """The reference's init zero-fills ~40 % of the tensors (conv2 / proj_out / ffn.3 / conv_out / adapters /
    cond_layers.*.1 and the norm biases), which makes outputs and 99.7 % of gradients vanish.  Replace every all-zero
    tensor by seeded noise (sigma = 1/sqrt(fan_in) for weights, 0.02 for vectors) so that no work is degenerate."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(state_dict):
        v = state_dict[k]
        if v.dtype.is_floating_point and v.numel() > 0 and float(v.abs().max()) == 0.0:
            if v.dim() >= 2:
                v = torch.randn(v.shape, generator=g, dtype=torch.float32) / math.sqrt(v[0].numel())
            else:
                v = 0.02 * torch.randn(v.shape, generator=g, dtype=torch.float32)
        out[k] = v.clone()
    return out

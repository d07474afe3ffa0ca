"""Tiny deterministic stand-ins for the vision model, used to test the diffusion/sampler host code on CPU
against the reference's Diffusion / NestedDiffusion (the real denoiser needs a GPU)."""
import torch
import torch.nn as nn


class StubUNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.input_channels = 3
        self.conditions = {"scale": 64.0}
        self.w = nn.Parameter(torch.tensor(0.35))

    def forward(self, x_t, times, lm_outputs, lm_mask, micros={}):
        n = x_t.shape[0]   # mixed-resolution batches: a level may see a prefix of the batch only
        t = (times[:n].float() / 1000.0).reshape(-1, 1, 1, 1)
        c = lm_outputs[:n].mean(dim=(1, 2)).reshape(-1, 1, 1, 1)
        return self.w * x_t + 0.1 * torch.tanh(3 * t) + 0.05 * c + 0.02 * torch.roll(x_t, 1, dims=-1)


class StubNestedUNet(StubUNet):
    def __init__(self):
        super().__init__()
        self.nest_ratio = [4]
        self.is_temporal = [False]

    def forward(self, x_t, times, lm_outputs, lm_mask, micros={}):
        return [StubUNet.forward(self, x, times, lm_outputs, lm_mask, micros) * (1.0 + 0.1 * i) for i, x in enumerate(x_t)]

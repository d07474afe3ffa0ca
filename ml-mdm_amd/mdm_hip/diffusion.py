"""Train-step and sampling call surface around the denoiser.

Host-side mirror of ``ml_mdm.diffusion`` (reference ml-mdm-matryoshka/ml_mdm/diffusion.py):
``DiffusionConfig`` (:29-50), ``Model`` (:53-87), ``Diffusion`` (:90-206),
``NestedDiffusionConfig`` (:214-248), ``NestedModel`` (:251-292), ``NestedDiffusion``
(:295-387) -- same names, same ``get_loss(sample)`` / ``sample(num_examples, sample,
image_side, device, **kw)`` signatures and return tuples, so ``trainer.train_batch``
(trainer.py:13-96) and the CLIs run against it unchanged.

``get_loss`` accepts optional ``time`` / ``noise_fn`` overrides so a parity test can feed
the same timesteps and noise to both implementations (the reference draws them from the
device RNG, samplers.py:236-241).

Loss-side arithmetic: on GPU tensors the noising ``x_t = sqrt(g) x + sqrt(1-g) eps``
is one kernel per scale, the image pyramid one ``mdm_avgpool`` per level, and prediction-space conversion + target
+ MSE + per-sample mean one differentiable kernel per scale (``ops.diffusion_loss``) -- 4 launches for the 64x64
pipeline where the torch formulation needs ~25.  CPU tensors (host-logic tests against the reference goldens) take
the torch formulation of the same formulas; nothing falls back from the GPU path.
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, samplers


@dataclass
class DiffusionConfig:
    sampler_config: samplers.SamplerConfig = field(default_factory=samplers.SamplerConfig)
    model_output_scale: float = 0
    use_vdm_loss_weights: bool = True


@dataclass
class NestedDiffusionConfig(DiffusionConfig):
    use_double_loss: bool = False
    multi_res_weights: str = None
    no_use_residual: bool = False
    use_random_interp: bool = False
    mixed_ratio: str = None
    random_downsample: bool = False
    average_downsample: bool = False
    mid_downsample: bool = False


class Model(nn.Module):
    """thin wrapper that owns the vision model (what DDP wraps, clis/train_parallel.py:148)."""

    def __init__(self, vision_model, diffusion_config: DiffusionConfig = None):
        super().__init__()
        self.diffusion_config = diffusion_config or DiffusionConfig()
        self._output_scale = self.diffusion_config.model_output_scale
        self.vision_model = vision_model
        self.sampler = None

    def set_sampler(self, sampler):
        self.sampler = sampler

    def load(self, vision_file):
        return self.vision_model.load(vision_file)

    def save(self, vision_file, other_items=None):
        self.vision_model.save(vision_file, other_items=other_items)

    @property
    def input_channels(self):
        return self.vision_model.input_channels

    def forward(self, x_t, times, lm_outputs, lm_mask, micros):
        out = self.vision_model(x_t, times, lm_outputs, lm_mask, micros)
        if self._output_scale != 0:
            out = torch.tanh(out / self._output_scale) * self._output_scale
        return out, torch.ones_like(out)


class Diffusion(nn.Module):
    def __init__(self, denoising_model, diffusion_config: DiffusionConfig):
        super().__init__()
        self.model = Model(denoising_model, diffusion_config)
        self.sampler = samplers.Sampler(diffusion_config.sampler_config)
        self.model.set_sampler(self.sampler)
        self._config = diffusion_config
        self.loss_fn = nn.MSELoss(reduction="none")
        # get_loss returns (loss, time, x_t, means, TARGETS, weights) like the reference; the fused GPU loss never
        # forms the target tensor, so it is built only on request (the trainer of this package does not ask)
        self.materialize_targets = True

    def get_model(self):
        return self.model.module if hasattr(self.model, "module") else self.model

    def to(self, device):
        self.model = self.model.to(device)
        self.sampler = self.sampler.to(device)
        return self

    def train(self, mode=True):
        self.model.train(mode)

    def eval(self):
        self.model.eval()
        self.sampler.eval()

    def get_micro_conditioning(self, sample):
        conditions = self.get_model().vision_model.conditions
        return {k: sample[k] for k in conditions if k in sample} if conditions is not None else {}

    def get_pred_for_training(self, x_t, pred, g):
        sc = self._config.sampler_config
        if sc.loss_target_type == sc.prediction_type:
            return pred
        x0, _ = self.sampler.get_x0_eps_from_pred(x_t, pred, g, sc.prediction_type)
        return self.sampler.get_pred_from_x0_xt(x_t, x0, g, sc.loss_target_type)

    def get_loss(self, sample, time=None, noise_fn=torch.randn_like):
        images, lm_outputs, lm_mask = sample["images"], sample["lm_outputs"], sample["lm_mask"]
        eps, g, g_last, weights, time = self.sampler.get_eps_time(images, time, noise_fn)
        if not self._config.use_vdm_loss_weights:
            weights = None
        if images.is_cuda:
            sc = self._config.sampler_config
            inv = 1.0 / sc.rescale_signal if sc.rescale_signal else 1.0
            x_t, eps = ops.noise_images(images.float(), g, eps.float(), inv_scale=inv)
            means, _ = self.model(x_t, time, lm_outputs, lm_mask, self.get_micro_conditioning(sample))
            # the reference builds x_t from the RESCALED images but the target from the raw ones (diffusion.py:153, 163):
            # the target term of the loss kernel gets scale 1 (they differ only for a V-prediction target with rescale_signal)
            loss = ops.diffusion_loss(means, x_t, images.float(), eps, g, sc.prediction_type, sc.loss_target_type, inv_scale=1.0)
            tgt = None
            if self.materialize_targets:
                tgt = self.sampler.get_prediction_targets(images, eps, g, g_last, sc.loss_target_type)
            return loss, time, x_t, means, tgt, weights
        x_t = self.sampler.get_xt(self.sampler.get_image_rescaled(images), eps, g)
        means, _ = self.model(x_t, time, lm_outputs, lm_mask, self.get_micro_conditioning(sample))
        tgt = self.sampler.get_prediction_targets(images, eps, g, g_last, self._config.sampler_config.loss_target_type)
        pred = self.get_pred_for_training(x_t, means, g)
        loss = self.loss_fn(pred, tgt).mean(axis=(1, 2, 3))
        return loss, time, x_t, means, tgt, weights

    def get_noise(self, num_examples, input_channels, image_side, device):
        return torch.randn(num_examples, input_channels, image_side, image_side).to(device)  # CPU RNG, as the reference (:177)

    def sample(self, num_examples, sample, image_side, device, **kwargs):
        self.eval()
        noise = self.get_noise(num_examples, self.get_model().input_channels, image_side, device)
        return self.sampler.sample(self.get_model(), noise, sample["lm_outputs"], sample["lm_mask"],
                                   self.get_micro_conditioning(sample), **kwargs)


class NestedModel(Model):
    def forward(self, x_t: List[torch.Tensor], times, lm_outputs, lm_mask, micros={}, mixed_ratio=None):
        if not self.diffusion_config.no_use_residual:
            # the reference's residual branch reads an undefined variable (diffusion.py:277-291) and both shipped
            # nested configs disable it (cc12m_256x256.yaml:24, cc12m_1024x1024.yaml:25)
            raise NotImplementedError("NestedModel requires no_use_residual=True")
        batch = x_t[0].size(0)
        if mixed_ratio is not None:  # run the higher resolutions on a prefix of the batch only
            x_t = [x[: int(m * x.size(0))] for x, m in zip(x_t, mixed_ratio)]
        p_t = self.vision_model(x_t, times, lm_outputs, lm_mask, micros)
        if mixed_ratio is not None:
            p_t = [torch.cat([p, p.new_zeros(batch - p.size(0), *p.shape[1:])], 0) for p in p_t]
        return p_t


class NestedDiffusion(Diffusion):
    def __init__(self, denoising_model, diffusion_config: NestedDiffusionConfig):
        nn.Module.__init__(self)
        self.model = NestedModel(denoising_model, diffusion_config)
        self.sampler = samplers.NestedSampler(diffusion_config.sampler_config)
        self.model.set_sampler(self.sampler)
        self._config = diffusion_config
        self.loss_fn = nn.MSELoss(reduction="none")
        self.materialize_targets = True
        self.mixed_ratio = None
        if diffusion_config.mixed_ratio:
            r = np.cumsum(np.asarray([float(x) for x in diffusion_config.mixed_ratio.split(":")]))
            self.mixed_ratio = r / r[-1]

    def get_loss(self, sample, time=None, noise_fn=torch.randn_like):
        images, lm_outputs, lm_mask = sample["images"], sample["lm_outputs"], sample["lm_mask"]
        micros = self.get_micro_conditioning(sample)
        vm = self.get_model().vision_model
        if any(vm.is_temporal):
            raise NotImplementedError("temporal nesting is not implemented")
        scales = vm.nest_ratio + [1]
        ratios = [scales[0] // s for s in scales]

        eps, g, g_last, weights, time = self.sampler.get_eps_time(images, time, noise_fn)
        if not self._config.use_vdm_loss_weights:
            weights = None
        # image pyramid by average pooling; the low-resolution noise is drawn independently (:333-356)
        hip = images.is_cuda
        pyr, noises = [images.float() if hip else images], [eps]
        for i in range(1, len(ratios)):
            rr = ratios[i] // ratios[i - 1]
            pyr.append(ops.avgpool(pyr[-1], rr) if hip else F.avg_pool2d(pyr[-1], rr))
            noises.append(noise_fn(pyr[-1]))
        g = self.sampler.get_gammas(g, scales, pyr)
        g_last = self.sampler.get_gammas(g_last, scales, pyr)
        if hip:
            return self._get_loss_hip(pyr, noises, g, g_last, scales, time, lm_outputs, lm_mask, micros, weights)

        x_t = self.sampler.get_xt(pyr, noises, g, scales)
        p_t = self.model(x_t, time, lm_outputs, lm_mask, micros, self.mixed_ratio)
        tgt = self.sampler.get_prediction_targets(pyr, noises, g, g_last, scales, self._config.sampler_config.loss_target_type)
        pred = [self.get_pred_for_training(x, p, gi) for x, p, gi in zip(x_t, p_t, g)]

        if self._config.multi_res_weights is not None:
            assert self._config.use_double_loss
            w = [float(v) for v in self._config.multi_res_weights.split(":")]
        else:
            w = [1.0] * len(x_t)
        loss = 0
        for i in range(len(x_t)):
            if i == 0 or self._config.use_double_loss:
                li = self.loss_fn(pred[i], tgt[i]).mean(axis=(1, 2, 3))
                if self.mixed_ratio is not None:
                    li = li / self.mixed_ratio[i]
                    li[int(self.mixed_ratio[i] * li.size(0)):] = 0
            else:
                li = pred[i].mean() * 0.0
            loss = loss + li * w[i]
        return loss, time, x_t[0], pred[0], tgt[0], weights

    def _get_loss_hip(self, pyr, noises, g, g_last, scales, time, lm_outputs, lm_mask, micros, weights):
        """GPU tensors: noising and loss are one kernel each per scale (see the module docstring)"""
        sc = self._config.sampler_config
        inv = [1.0 if sc.schedule_shifted else (1.0 / s if s else 1.0) for s in scales]   # NestedSampler._signal
        x_t = [ops.noise_images(x, gi, e.float(), inv_scale=iv)[0] for x, gi, e, iv in zip(pyr, g, noises, inv)]
        p_t = self.model(x_t, time, lm_outputs, lm_mask, micros, self.mixed_ratio)
        if self._config.multi_res_weights is not None:
            assert self._config.use_double_loss
            w = [float(v) for v in self._config.multi_res_weights.split(":")]
        else:
            w = [1.0] * len(x_t)
        loss = 0
        for i in range(len(x_t)):
            if i == 0 or self._config.use_double_loss:
                li = ops.diffusion_loss(p_t[i], x_t[i], pyr[i], noises[i].float(), g[i], sc.prediction_type,
                                        sc.loss_target_type, inv_scale=inv[i])
                if self.mixed_ratio is not None:
                    li = li / self.mixed_ratio[i]
                    li = torch.cat([li[: int(self.mixed_ratio[i] * li.size(0))], li.new_zeros(li.size(0) - int(self.mixed_ratio[i] * li.size(0)))])
            else:
                li = p_t[i].mean() * 0.0
            loss = loss + li * w[i]
        pred0 = tgt0 = None
        if self.materialize_targets:
            tgt0 = samplers.Sampler.get_prediction_targets(self.sampler, self.sampler._signal(pyr[0], scales[0]), noises[0],
                                                           g[0], g_last[0], sc.loss_target_type)
            pred0 = self.get_pred_for_training(x_t[0], p_t[0], g[0])
        return loss, time, x_t[0], pred0, tgt0, weights

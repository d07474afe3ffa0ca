"""Noise schedules and the DDPM / DDIM sampling update around the denoiser.

Host-side mirror of ``ml_mdm.samplers`` (reference ml-mdm-matryoshka/ml_mdm/samplers.py):
``SamplerConfig`` (:67-118), the schedule families (:126-165), ``Sampler`` (:177-609) and
``NestedSampler`` (:612-793) with the same public method names, so the reference's
``Diffusion`` / CLIs can drive it.

Per-pixel arithmetic: on GPU tensors one reverse step --
guidance combine, v/eps -> x0, clip / dynamic threshold, DDPM posterior or DDIM(eta)
update, noise -- is ONE kernel per scale (``ops.sampler_step`` -> ``mdm_sampler_step``);
there is no fallback from that path (a missing library raises).  The same formulas written
with torch ops serve CPU tensors: that is what the host-logic tests drive against the
reference's golden outputs (tests/test_diffusion_host.py), and what a custom ``clip_fn``
gets.  One deliberate difference from the reference: gammas are per-sample ``[B, 1, 1, 1]``
tensors, broadcast, instead of being materialised at full image size (:196-199).
"""
import math
from dataclasses import dataclass
from enum import Enum

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class ScheduleType(Enum):
    COSINE = 0
    DDPM = 1
    DEEPFLOYD = 2


class PredictionType(Enum):
    DDPM = 0
    DDIM = 1
    V_PREDICTION = 2


class ThresholdType(Enum):
    NONE = 0
    CLIP = 1
    DYNAMIC = 2
    DYNAMIC_IF = 3


def _enum(cls, v):
    return v if isinstance(v, cls) else cls[str(v).upper()]


@dataclass
class SamplerConfig:
    num_diffusion_steps: int = 32
    reproject_signal: bool = False
    schedule_type: ScheduleType = ScheduleType.DDPM
    prediction_type: PredictionType = PredictionType.DDPM
    loss_target_type: PredictionType = None
    beta_start: float = 0.0001
    beta_end: float = 0.02
    threshold_function: ThresholdType = ThresholdType.CLIP
    rescale_schedule: float = 1.0
    rescale_signal: float = None
    schedule_shifted: bool = False
    schedule_shifted_power: float = 1

    def __post_init__(self):
        self.schedule_type = _enum(ScheduleType, self.schedule_type)
        self.prediction_type = _enum(PredictionType, self.prediction_type)
        if self.loss_target_type is not None:
            self.loss_target_type = _enum(PredictionType, self.loss_target_type)
        self.threshold_function = _enum(ThresholdType, self.threshold_function)


def gammas_cosine(n, logsnr_min=-5.0, logsnr_max=5.0):
    """reference :126-136 (progressive-distillation cosine log-SNR schedule); gamma_0 = 1."""
    t = np.linspace(0.0, 1.0, num=n)
    b = np.arctan(np.exp(-0.5 * logsnr_max))
    a = np.arctan(np.exp(-0.5 * logsnr_min)) - b
    logsnr = -2.0 * np.log(np.tan(a * t + b))
    return np.concatenate(([1.0], 1.0 / (1.0 + np.exp(-logsnr))))


def gammas_linear_beta(n, beta_start, beta_end):
    """reference :139-146 (Ho et al. linear betas); gamma_t = prod_{s<=t} (1 - beta_s), beta_0 = 0."""
    betas = np.concatenate(([0.0], np.linspace(beta_start, beta_end, num=n)))
    return np.exp(np.cumsum(np.log(1.0 - betas)))


def gammas_squaredcos_cap_v2(n):
    """reference :149-165 (DeepFloyd / diffusers squaredcos_cap_v2), betas capped at 0.999."""
    bar = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = np.asarray([0.0] + [min(1 - bar((i + 1) / n) / bar(i / n), 0.999) for i in range(n)])
    return np.exp(np.cumsum(np.log(1.0 - betas)))


def _b(v):
    """[B] -> [B, 1, 1, 1]"""
    return v.reshape(-1, 1, 1, 1)


class Sampler(nn.Module):
    def __init__(self, sampler_config: SamplerConfig):
        super().__init__()
        self._config = cfg = sampler_config
        self.n_steps = cfg.num_diffusion_steps
        if cfg.schedule_type == ScheduleType.COSINE:
            g = gammas_cosine(self.n_steps)
        elif cfg.schedule_type == ScheduleType.DDPM:
            g = gammas_linear_beta(self.n_steps, cfg.beta_start, cfg.beta_end)
        elif cfg.schedule_type == ScheduleType.DEEPFLOYD:
            g = gammas_squaredcos_cap_v2(self.n_steps)
        else:
            raise ValueError("unknown schedule")
        self.register_buffer("_gammas", torch.tensor(g).float())
        gammas = self.get_schedule_shifted(self._gammas.clone(), cfg.rescale_schedule)
        gt, gl = gammas[2:], gammas[1:-1]
        w = gl * (1 - gt) / (1 - gl) / gt - 1  # VDM weights (:223-228)
        self.register_buffer("gammas", gammas)
        self.register_buffer("vdm_loss_weights", torch.cat([w[:1], w[:1], w]))
        if cfg.loss_target_type is None:
            cfg.loss_target_type = cfg.prediction_type
        self.device_rng = None   # ops.DeviceRng: draw the sampling noise inside the step kernel (see use_device_rng)

    def use_device_rng(self, seed: int, device):
        """Draw the ancestral-sampling noise INSIDE the step kernel from the library's counter-based generator
        (replayable on the host, oracle/philox_ref.py) instead of ``torch.randn_like`` -- one launch less per scale."""
        self.device_rng = ops.DeviceRng(seed, device)
        return self

    # ---- schedule access -------------------------------------------------------------
    def read_gamma(self, time, image=None):
        return _b(self.gammas[time])

    def get_schedule_shifted(self, gammas, scale_factor=None):
        """SNR' = SNR / s^p (:255-264)."""
        if scale_factor is not None and scale_factor > 1:
            s = scale_factor ** self._config.schedule_shifted_power
            snr = gammas / (1 - gammas)
            gammas = 1 / (1 + s / snr)
        return gammas

    def get_image_rescaled(self, images, scale_factor=None):
        s = self._config.rescale_signal if scale_factor is None else scale_factor
        return images / s if s else images

    # ---- training-side helpers (:233-279, 347-390) --------------------------------------
    def get_eps_time(self, images, time=None, noise_fn=torch.randn_like):
        B = images.shape[0]
        if time is None:
            time = torch.randint(0, self.n_steps, (B,), device=images.device)
        else:  # scalar or per-sample tensor
            time = torch.as_tensor(time, device=images.device) * torch.ones(B, dtype=torch.long, device=images.device)
        return noise_fn(images), self.read_gamma(time + 1), self.read_gamma(time), self.vdm_loss_weights[time + 1], time

    def get_xt(self, images, eps, g):
        return g.sqrt() * images + (1 - g).sqrt() * eps

    def get_prediction_targets(self, images, eps, g, g_last, prediction_type=None):
        pt = prediction_type or self._config.loss_target_type
        if pt in (PredictionType.DDPM, PredictionType.DDIM):
            return eps
        if pt == PredictionType.V_PREDICTION:
            return g.sqrt() * eps - (1 - g).sqrt() * images
        raise ValueError("unsupported prediction type")

    def get_x0_eps_from_pred(self, x_t, pred, g, prediction_type=None, clip_fn=None, return_eps=True):
        pt = prediction_type or self._config.prediction_type
        if pt in (PredictionType.DDPM, PredictionType.DDIM):
            x0 = (x_t - pred * (1 - g).sqrt()) / g.sqrt()
        elif pt == PredictionType.V_PREDICTION:
            x0 = x_t * g.sqrt() - pred * (1 - g).sqrt()
        else:
            raise ValueError("unsupported prediction type")
        if clip_fn is not None:
            x0 = clip_fn(x0)
        if not return_eps:
            return x0
        return x0, (x_t - x0 * g.sqrt()) / (1 - g).sqrt()

    def get_pred_from_x0_xt(self, x_t, x0, g, prediction_type=None):
        pt = prediction_type or self._config.prediction_type
        if pt in (PredictionType.DDPM, PredictionType.DDIM):
            return (x_t - x0 * g.sqrt()) / (1 - g).sqrt()
        if pt == PredictionType.V_PREDICTION:
            return (g.sqrt() * x_t - x0) / (1 - g).sqrt()
        raise ValueError("unsupported prediction type")

    # ---- one reverse step (:281-345) -------------------------------------------------------
    def get_prediction_xt_last(self, x_t, pred, g, g_last, prediction_type=None, clip_fn=None, need_noise=False,
                               ddim_eta=None, input_noise=None, image_scale=None, return_eps=True, pred_uncond=None,
                               guidance_scale=1):
        if x_t.is_cuda and (clip_fn is None or clip_fn == self.clip_sample):
            return self._xt_last_hip(x_t, pred, g, g_last, prediction_type, clip_fn is not None, need_noise, ddim_eta,
                                     input_noise, image_scale, return_eps, pred_uncond, guidance_scale)
        if pred_uncond is not None:
            pred = pred_uncond + guidance_scale * (pred - pred_uncond)
        alpha = g / g_last
        beta = 1 - alpha
        beta_tilde = beta * (1 - g_last) / (1 - g)
        x0 = self.get_x0_eps_from_pred(x_t, pred, g, prediction_type=prediction_type, return_eps=False)
        scale = 1 if image_scale is None else image_scale
        x0 = torch.clip(x0, -scale, scale) / scale if clip_fn is None else clip_fn(x0, scale)
        if ddim_eta is None:  # ancestral DDPM posterior mean
            x_last = x0 * beta * g_last.sqrt() / (1 - g) + x_t * alpha.sqrt() * (1 - g_last) / (1 - g)
        else:
            eps = (x_t - x0 * g.sqrt()) / (1 - g).sqrt()
            if ddim_eta > 0:
                beta_tilde = (ddim_eta ** 2) * beta_tilde
                x_last = x0 * g_last.sqrt() + eps * (1 - g_last - beta_tilde).sqrt()
            else:
                need_noise = False
                x_last = x0 * g_last.sqrt() + eps * (1 - g_last).sqrt()
        if need_noise:
            noise = torch.randn_like(x_last) if input_noise is None else input_noise
            x_last = x_last + beta_tilde.sqrt() * noise
        eps = (x_last - g_last.sqrt() * x0) / (1 - g_last).sqrt() if return_eps else None
        return x0, x_last, eps

    def _xt_last_hip(self, x_t, pred, g, g_last, prediction_type, use_clip_sample, need_noise, ddim_eta, input_noise,
                     image_scale, return_eps, pred_uncond, guidance_scale):
        """the same update as ONE kernel (two for dynamic thresholding: the quantile sits between them)"""
        pt = prediction_type or self._config.prediction_type
        scale = 1 if image_scale is None else image_scale
        x_t, pred = x_t.float(), pred.float()
        kw = dict(prediction_type=pt, ddim_eta=ddim_eta, image_scale=scale, guidance_scale=guidance_scale,
                  pred_uncond=None if pred_uncond is None else pred_uncond.float())
        fn = self._config.threshold_function if use_clip_sample else None
        thr, clip = None, "NONE"
        if fn in (ThresholdType.DYNAMIC, ThresholdType.DYNAMIC_IF):
            ratio, vmax = (0.995, 100) if fn == ThresholdType.DYNAMIC else (0.95, 1.5)
            x0s, _ = ops.sampler_step(x_t, pred, g, g_last, clip="X0_ONLY", **kw)
            thr = torch.quantile(x0s.reshape(x0s.shape[0], -1).abs(), ratio, dim=1).clamp(min=1, max=vmax)
            clip = "DYNAMIC"
        elif fn == ThresholdType.CLIP:
            clip = "CLIP"
        elif not use_clip_sample:
            # clip_fn=None in the reference: clamp(x0, -s, s) / s == CLIP on x0 / s with unit scale ... only for s == 1
            if scale != 1:
                raise NotImplementedError("clip_fn=None with image_scale != 1 on the HIP path")
            clip = "CLIP"
        noisy = need_noise and not (ddim_eta is not None and ddim_eta <= 0)
        noise = input_noise
        if noisy and noise is None and self.device_rng is None:
            noise = torch.randn_like(x_t)   # torch's generator, like the reference (:340)
        x0, x_last = ops.sampler_step(x_t, pred, g, g_last, need_noise=noisy, noise=noise, rng=self.device_rng, clip=clip,
                                      thr=thr, **kw)
        if noisy and noise is None:
            self.device_rng.advance(x_t.numel())
        eps = (x_last - _b(g_last).sqrt() * x0) / (1 - _b(g_last)).sqrt() if return_eps else None
        return x0, x_last, eps

    def _threshold_sample(self, sample, ratio=0.995, max_value=100):
        """Imagen dynamic thresholding (:461-498)."""
        shape, dtype = sample.shape, sample.dtype
        flat = sample.float().reshape(shape[0], -1)
        s = torch.quantile(flat.abs(), ratio, dim=1).clamp(min=1, max=max_value).unsqueeze(1)
        return (torch.clamp(flat, -s, s) / s).reshape(shape).to(dtype)

    def clip_sample(self, pred_x0, image_scale=1):
        s, fn = image_scale, self._config.threshold_function
        if fn == ThresholdType.CLIP:
            return (pred_x0 * s).clip(-1, 1) / s
        if fn == ThresholdType.DYNAMIC:
            return self._threshold_sample(pred_x0 * s, 0.995, 100) / s
        if fn == ThresholdType.DYNAMIC_IF:
            return self._threshold_sample(pred_x0 * s, 0.95, 1.5) / s
        return pred_x0

    def forward_model(self, model, x_t, t, lm_outputs, lm_mask, micros={}, guidance_scale=1):
        """classifier-free guidance doubles the batch: [uncond | cond] (:435-459)."""
        pc, pu, extras = self._forward_model_raw(model, x_t, t, lm_outputs, lm_mask, micros, guidance_scale)
        return (pc if pu is None else pu + guidance_scale * (pc - pu)), extras

    def _forward_model_raw(self, model, x_t, t, lm_outputs, lm_mask, micros, guidance_scale):
        """-> (conditional prediction, unconditional prediction or None, extras): the guidance combine itself is
        folded into the step kernel"""
        if guidance_scale != 1:
            assert x_t.shape[0] * 2 == lm_outputs.shape[0]
            pred, extras = model(torch.cat([x_t] * 2), torch.cat([t, t]), lm_outputs, lm_mask, micros=micros)
            pu, pc = pred.chunk(2)
            return pc, pu, extras.chunk(2)[1]
        pred, extras = model(x_t, t, lm_outputs, lm_mask, micros)
        return pred, None, extras

    def get_xt_minus_1(self, model, time_step, x_t, lm_outputs, lm_mask, micros={}, time_step_last=None,
                       guidance_scale=1, ddim_eta=None, return_details=False):
        ones = torch.ones(x_t.shape[0], dtype=torch.long, device=self.gammas.device)
        last = time_step - 1 if time_step_last is None else time_step_last
        t, s = ones * time_step, ones * last
        g, g_last = self.read_gamma(t), self.read_gamma(s)
        pc, pu, _ = self._forward_model_raw(model, x_t, t - 1, lm_outputs, lm_mask, micros, guidance_scale)  # model sees t-1 (:415)
        x0, x_s, _ = self.get_prediction_xt_last(
            x_t, pc, g, g_last, prediction_type=self._config.prediction_type, need_noise=bool(last != 0),
            ddim_eta=ddim_eta, clip_fn=self.clip_sample, image_scale=self._config.rescale_signal, return_eps=False,
            pred_uncond=pu, guidance_scale=guidance_scale)
        return (x0, x_s, (g, g_last)) if return_details else x_s

    # ---- the sampling loop (:510-609) ----------------------------------------------------------
    def set_timesteps(self, num_inference_steps=250):
        ratio = (self._config.num_diffusion_steps + 1) / (num_inference_steps + 1)
        return (np.arange(0, num_inference_steps + 1) * ratio).round()[::-1].copy().astype(np.int64)

    def sample(self, *args, **kwargs):
        gen = self._sample(*args, **kwargs)
        return gen if kwargs.get("yield_output", False) else next(gen)

    def _sample(self, model, x_t, lm_outputs, lm_mask, micros, return_sequence=False, use_beta_tilde=False, t=-1,
                num_inference_steps=2000, ddim_eta=None, guidance_scale=1, resample_steps=False, disable_bar=True,
                yield_output=False, **post_args):
        assert not (yield_output and return_sequence)
        if not resample_steps:
            num_inference_steps = self.n_steps
        steps = torch.from_numpy(self.set_timesteps(num_inference_steps)).to(self.gammas.device)
        if t > -1:
            steps = steps[steps <= t]
        seq = [x_t] if return_sequence else []
        x0 = extra = None
        for i, ts in enumerate(steps[:-1]):
            x0, x_t, extra = self.get_xt_minus_1(
                model, ts, x_t, lm_outputs, lm_mask, micros, time_step_last=steps[i + 1] if resample_steps else None,
                guidance_scale=guidance_scale, ddim_eta=ddim_eta, return_details=True)
            if yield_output:
                yield self._postprocess(x_t, x0, extra, **post_args)
            if return_sequence:
                seq.append(self._postprocess(x_t))
        if return_sequence:
            seq[-1] = torch.clip(seq[-1], -1, 1)
            yield seq
        else:
            yield self._postprocess(x_t, x0, extra, clip=True, **post_args)

    def _postprocess(self, x_t, x0=None, extra=None, yield_full=False, clip=False, image_scale=None, **unused):
        scale = self._config.rescale_signal if image_scale is None else image_scale
        if scale:
            x0 = x0 * scale if x0 is not None else x0
            x_t = x_t * scale
        if clip:
            x_t = torch.clip(x_t, -1, 1)
        return (x0, x_t, extra) if yield_full else x_t


class NestedSampler(Sampler):
    """Multi-resolution variant: lists of images, highest resolution first (reference :612-793)."""

    def get_gammas(self, gamma, scales, images=None):
        if not self._config.schedule_shifted:
            return [gamma for _ in scales]
        return [self.get_schedule_shifted(gamma, s) for s in scales]

    def _signal(self, x, s):
        return x if self._config.schedule_shifted else self.get_image_rescaled(x, s)

    def get_xt(self, x0, eps, g, scales):
        return [Sampler.get_xt(self, self._signal(x, s), e, gi) for x, s, e, gi in zip(x0, scales, eps, g)]

    def get_prediction_targets(self, x0, eps, g, g_last, scales, prediction_type=None):
        return [Sampler.get_prediction_targets(self, self._signal(x, s), e, gi, gl, prediction_type)
                for x, s, e, gi, gl in zip(x0, scales, eps, g, g_last)]

    def forward_model(self, model, x_t, t, lm_outputs, lm_mask, micros={}, guidance_scale=1):
        pcs, pus = self._forward_model_raw(model, x_t, t, lm_outputs, lm_mask, micros, guidance_scale)
        return [pc if pu is None else pu + guidance_scale * (pc - pu) for pc, pu in zip(pcs, pus)]

    def _forward_model_raw(self, model, x_t, t, lm_outputs, lm_mask, micros, guidance_scale):
        """-> (conditional predictions, unconditional predictions or Nones), one entry per scale (:777-790)"""
        if guidance_scale != 1:
            assert x_t[0].shape[0] * 2 == lm_outputs.shape[0]
            p_t = model([torch.cat([x] * 2) for x in x_t], torch.cat([t] * 2), lm_outputs, lm_mask, micros)
            halves = [p.chunk(2) for p in p_t]
            return [h[1] for h in halves], [h[0] for h in halves]
        p_t = model(x_t, t, lm_outputs, lm_mask, micros)
        return list(p_t), [None] * len(p_t)

    def get_xt_minus_1(self, model, time_step, x_t, lm_outputs, lm_mask, micros={}, time_step_last=None,
                       guidance_scale=1, ddim_eta=None, return_details=False):
        scales = model.vision_model.nest_ratio + [1]
        if isinstance(x_t, torch.Tensor):  # first step: draw independent noise at every lower resolution (:669-676)
            pyramid = [x_t]
            for s in scales[1:]:
                r = scales[0] // s
                pyramid.append(torch.randn_like(F.avg_pool2d(x_t, r)))
            x_t = pyramid
        ones = torch.ones(x_t[0].shape[0], dtype=torch.long, device=self.gammas.device)
        t = ones * time_step
        s = t - 1 if time_step_last is None else ones * time_step_last
        g_t = self.get_gammas(self.read_gamma(t), scales)
        g_s = self.get_gammas(self.read_gamma(s), scales)
        pcs, pus = self._forward_model_raw(model, x_t, t - 1, lm_outputs, lm_mask, micros, guidance_scale)
        x0, x_s = [], []
        for x, pc, pu, g, gl, sc in zip(x_t, pcs, pus, g_t, g_s, scales):
            a, b, _ = self.get_prediction_xt_last(
                x, pc, g, gl, prediction_type=self._config.prediction_type, need_noise=bool(time_step != 1),
                ddim_eta=ddim_eta, clip_fn=self.clip_sample, image_scale=sc if not self._config.schedule_shifted else 1,
                return_eps=False, pred_uncond=pu, guidance_scale=guidance_scale)
            x0.append(a)
            x_s.append(b)
        return (x0, x_s, (g_t[-1], g_s[-1])) if return_details else x_s

    def _postprocess(self, x_t, x0=None, extra=None, yield_full=False, clip=False, output_inner=False, **unused):
        scales = [1 if self._config.schedule_shifted else x.size(-1) / x_t[-1].size(-1) for x in x_t]
        one = lambda i: Sampler._postprocess(self, x_t[i], x0[i] if x0 is not None else None, extra,
                                             yield_full=yield_full, clip=clip, image_scale=scales[i], **unused)
        out = one(0)
        if not output_inner:
            return out
        outs = [out] + [one(i) for i in range(1, len(x_t))]
        up = lambda x, size: F.interpolate(x, size, mode="bilinear")
        if not yield_full:
            return torch.cat([up(o, outs[0].size(-1)) for o in outs[::-1]], -1)
        a, b, e = zip(*outs)
        return (torch.cat([up(v, a[0].size(-1)) for v in a[::-1]], -1),
                torch.cat([up(v, b[0].size(-1)) for v in b[::-1]], -1), e[-1])

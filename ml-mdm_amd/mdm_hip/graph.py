"""hipGraph replay for sampling.

Sampling is hundreds of strictly sequential denoiser calls on fixed shapes (reference samplers.py:516-578).  At small
batch the ~1500 kernel launches of one call are host-bound (Python + launch, whatever the GPU needs), so:

``GraphedDenoiser``   the model forward alone as a graph (drop-in wrapper around the vision model: the eager sampler
                      keeps driving it).
``GraphedSampler``    ONE WHOLE DENOISE ITERATION as a graph -- per-step schedule lookup, denoiser forward (with the
                      text conditioning hoisted out of the loop: it does not depend on t), the fused reverse-step
                      kernel of every scale (guidance combine, x0, clip, DDPM / DDIM update, in-kernel noise), the
                      step-counter and RNG advance -- replayed ``num_inference_steps`` times with no host work in
                      between.  All per-step values live in device tables indexed by a device-side counter.

Inference only (no autograd through a replay); parameters must not change between replays -- call ``reset()`` after
loading a checkpoint (the packed kernel-layout weights baked into the graph would be stale otherwise).
"""
import torch
import torch.nn as nn

from . import ops
from .samplers import NestedSampler, ThresholdType


class GraphedDenoiser(nn.Module):
    def __init__(self, model: nn.Module, warmup: int = 2):
        super().__init__()
        self.model = model
        self._warmup = warmup
        self._graphs = {}

    # attributes the diffusion / sampler code reads from the vision model
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("model"), name)

    def reset(self):
        self._graphs.clear()

    @staticmethod
    def _sig(x_t, times, cond, mask):
        xs = x_t if isinstance(x_t, (list, tuple)) else [x_t]
        return (tuple(tuple(x.shape) for x in xs), isinstance(x_t, (list, tuple)), tuple(times.shape), times.dtype,
                None if cond is None else tuple(cond.shape), None if mask is None else tuple(mask.shape),
                torch.is_autocast_enabled(), xs[0].device.index)

    def forward(self, x_t, times, conditioning=None, cond_mask=None, micros={}):
        if torch.is_grad_enabled() or self.model.training or micros:
            return self.model(x_t, times, conditioning, cond_mask, micros)
        key = self._sig(x_t, times, conditioning, cond_mask)
        ent = self._graphs.get(key)
        is_list = isinstance(x_t, (list, tuple))
        xs = list(x_t) if is_list else [x_t]
        if ent is None:
            static_x = [x.clone() for x in xs]
            static_t = times.clone()
            static_c = conditioning.clone() if conditioning is not None else None
            static_m = cond_mask.clone() if cond_mask is not None else None
            arg_x = static_x if is_list else static_x[0]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):   # warm-up: packs weights, sets kernel attributes, sizes the allocator
                for _ in range(self._warmup):
                    self.model(arg_x, static_t, static_c, static_m, {})
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.model(arg_x, static_t, static_c, static_m, {})
            ent = (graph, static_x, static_t, static_c, static_m, static_out)
            self._graphs[key] = ent
        graph, static_x, static_t, static_c, static_m, static_out = ent
        for s, x in zip(static_x, xs):
            s.copy_(x)
        static_t.copy_(times)
        if static_c is not None:
            static_c.copy_(conditioning)
        if static_m is not None:
            static_m.copy_(cond_mask)
        graph.replay()
        if isinstance(static_out, (list, tuple)):
            return [o.clone() for o in static_out]
        return static_out.clone()


class GraphedSampler:
    """``Diffusion.sample`` / ``NestedDiffusion.sample`` with one hipGraph replay per denoise step.

    Semantics are those of ``Sampler._sample`` with ``resample_steps=True`` (reference samplers.py:516-609, 655-713):
    DDPM (``ddim_eta=None``) or DDIM(eta), classifier-free guidance, CLIP / NONE / DYNAMIC / DYNAMIC_IF thresholding (the
    dynamic ones: x0 kernel -> ``torch.quantile`` -> update kernel, all inside the graph).  The ancestral noise comes
    from the library's counter-based generator (``ops.DeviceRng``, replayable on the host); the START noise is drawn
    like the eager path (CPU generator for the top scale, reference diffusion.py:177), or passed in."""

    def __init__(self, pipeline, warmup: int = 2, seed: int = 0):
        self.pipe = pipeline
        self.sampler = pipeline.sampler
        self._warmup = warmup
        self._seed = seed
        self._graphs = {}
        fn = self.sampler._config.threshold_function
        if fn not in (ThresholdType.CLIP, ThresholdType.NONE, ThresholdType.DYNAMIC, ThresholdType.DYNAMIC_IF):
            raise NotImplementedError("GraphedSampler: unknown threshold function %r" % (fn,))

    def reset(self):
        self._graphs.clear()

    # ---- host-side schedule ------------------------------------------------------------------------------
    def _tables(self, n_steps, device):
        steps = self.sampler.set_timesteps(n_steps)                 # n+1 entries, descending, last = 0
        t, s = steps[:-1], steps[1:]
        nested = isinstance(self.sampler, NestedSampler)
        # the last step adds no noise: Sampler tests `last != 0` (:421), NestedSampler `time_step != 1` (:700)
        gate = (t != 1) if nested else (s != 0)
        mk = lambda a, dt: torch.as_tensor(a.copy()).to(device=device, dtype=dt)
        return mk(t, torch.long), mk(s, torch.long), mk(gate.astype("float32"), torch.float32)

    def _build(self, key, xs, cond, mask, n_steps, ddim_eta, guidance, micros):
        smp, cfg = self.sampler, self.sampler._config
        model = self.pipe.get_model()
        vm = model.vision_model
        dev = xs[0].device
        nested = isinstance(smp, NestedSampler)
        B = xs[0].shape[0]
        scales = (vm.nest_ratio + [1]) if nested else [1]
        tab_t, tab_s, tab_gate = self._tables(n_steps, dev)
        idx = torch.zeros(1, dtype=torch.long, device=dev)
        rng = ops.DeviceRng(self._seed, dev)
        x_static = [x.clone() for x in xs]
        micros_s = {k: v.clone() for k, v in micros.items()}   # micro-conditioning ([B] per key, diffusion.py:136-141): static inputs
        # the text path (lm_proj, masked mean, cond_emb) does not depend on t: computed once per sample() call into
        # static buffers, outside the per-step graph
        ce, cs, cm = vm.forward_conditioning(cond, mask)
        ce_s, cs_s, cm_s = ce.clone(), cs.clone(), (cm.clone() if cm is not None else None)
        out_scale = model._output_scale
        fn = cfg.threshold_function
        clip = {ThresholdType.CLIP: "CLIP", ThresholdType.NONE: "NONE"}.get(fn, "DYNAMIC")
        # Imagen dynamic thresholding (samplers.py:461-498): a per-sample quantile of |x0| sits between two launches of
        # the step kernel (x0 only, then the update with the threshold) -- torch.quantile is a sort + lerp on the
        # device, so it is captured with the rest
        dyn = {ThresholdType.DYNAMIC: (0.995, 100.0), ThresholdType.DYNAMIC_IF: (0.95, 1.5)}.get(fn)
        noisy = not (ddim_eta is not None and ddim_eta <= 0)

        def body():
            t = tab_t.index_select(0, idx)
            s = tab_s.index_select(0, idx)
            gate = tab_gate.index_select(0, idx)
            g_t, g_s = smp.gammas.index_select(0, t).expand(B), smp.gammas.index_select(0, s).expand(B)
            if nested:
                g_t, g_s = smp.get_gammas(g_t, scales), smp.get_gammas(g_s, scales)
            else:
                g_t, g_s = [g_t], [g_s]
            times = (t - 1).expand(B)
            if guidance != 1:
                xin = [torch.cat([x, x]) for x in x_static]
                tin = torch.cat([times, times])
            else:
                xin, tin = x_static, times
            preds = vm.forward_denoising(xin if nested else xin[0], tin, ce_s, cs_s, cm_s, micros_s)
            preds = list(preds) if nested else [preds]
            if out_scale != 0:
                preds = [torch.tanh(p / out_scale) * out_scale for p in preds]
            for i, (x, p, sc) in enumerate(zip(x_static, preds, scales)):
                pu = None
                if guidance != 1:
                    pu, p = p.chunk(2)
                img_scale = (sc if not cfg.schedule_shifted else 1) if nested else (cfg.rescale_signal or 1)
                thr = None
                if dyn is not None:
                    x0s, _ = ops.sampler_step(x, p, g_t[i], g_s[i], cfg.prediction_type, ddim_eta=ddim_eta, clip="X0_ONLY",
                                              image_scale=img_scale, pred_uncond=pu, guidance_scale=guidance)
                    thr = torch.quantile(x0s.reshape(B, -1).abs(), dyn[0], dim=1).clamp(min=1, max=dyn[1])
                _, x_last = ops.sampler_step(x, p, g_t[i], g_s[i], cfg.prediction_type, ddim_eta=ddim_eta, need_noise=noisy,
                                             rng=rng, clip=clip, thr=thr, image_scale=img_scale, pred_uncond=pu,
                                             guidance_scale=guidance, noise_gate=gate)
                if noisy:
                    rng.advance(x.numel())   # same draw order as the eager sampler with use_device_rng()
                x.copy_(x_last)
            idx.add_(1)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):   # warm-up: packs weights, sets kernel attributes, sizes the allocator
            for _ in range(self._warmup):
                idx.zero_()           # every warm-up pass runs step 0 (a 1-step schedule has no step 1 to index)
                body()
            # ... and leaves no trace: the first replay starts at step 0 with the generator at (seed, 0), like the
            # eager sampler after use_device_rng(seed)
            idx.zero_()
            rng.state.copy_(torch.tensor([self._seed & (2**63 - 1), 0], dtype=torch.int64))
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        # everything the graph reads by raw pointer must outlive it: the schedule tables too (they are locals of this
        # function; once freed, a later allocation reuses their memory and the replayed index_select reads garbage
        # indices -> out-of-bounds gather)
        ent = dict(graph=graph, x=x_static, ce=ce_s, cs=cs_s, cm=cm_s, idx=idx, rng=rng, n=n_steps, micros=micros_s,
                   keep=(tab_t, tab_s, tab_gate))
        self._graphs[key] = ent
        return ent

    @torch.no_grad()
    def sample(self, num_examples, sample, image_side, device, num_inference_steps=50, ddim_eta=None, guidance_scale=1,
               start_noise=None, seed=None):
        """-> images like ``Diffusion.sample(..., resample_steps=True, num_inference_steps=n, ddim_eta=eta,
        guidance_scale=w)``.  ``start_noise`` (tensor, or hi->lo list for a nested model) replaces the drawn x_T."""
        self.pipe.eval()
        smp = self.sampler
        nested = isinstance(smp, NestedSampler)
        model = self.pipe.get_model()
        if start_noise is None:
            x = self.pipe.get_noise(num_examples, model.input_channels, image_side, device)
            xs = [x]
            if nested:   # independent noise at every lower resolution (:669-676)
                scales = model.vision_model.nest_ratio + [1]
                xs += [torch.randn(num_examples, x.shape[1], image_side * s // scales[0], image_side * s // scales[0], device=x.device)
                       for s in scales[1:]]
        else:
            xs = [t.to(device).float() for t in (start_noise if isinstance(start_noise, (list, tuple)) else [start_noise])]
        cond, mask = sample["lm_outputs"], sample["lm_mask"]
        if guidance_scale != 1:
            assert xs[0].shape[0] * 2 == cond.shape[0], "classifier-free guidance: lm_outputs = [uncond | cond]"
        # micro-conditioning the sample carries (scale, watermark_score, ...), as Diffusion.sample passes it on
        # (reference diffusion.py:194-196); one value per row of the model's batch (2B under guidance)
        micros = {k: torch.as_tensor(v, device=xs[0].device).reshape(-1).float()
                  for k, v in self.pipe.get_micro_conditioning(sample).items()}
        key = (tuple(tuple(x.shape) for x in xs), tuple(cond.shape), int(num_inference_steps), ddim_eta, float(guidance_scale),
               torch.is_autocast_enabled(), torch.get_autocast_gpu_dtype() if torch.is_autocast_enabled() else None,
               ops.fp32_split_enabled(),   # a captured graph keeps the arithmetic it was captured with
               tuple(sorted((k, tuple(v.shape)) for k, v in micros.items())))
        ent = self._graphs.get(key) or self._build(key, xs, cond, mask, int(num_inference_steps), ddim_eta, float(guidance_scale), micros)
        for sx, x in zip(ent["x"], xs):
            sx.copy_(x)
        for k, v in micros.items():
            ent["micros"][k].copy_(v)
        ce, cs, cm = model.vision_model.forward_conditioning(cond, mask)
        ent["ce"].copy_(ce)
        ent["cs"].copy_(cs)
        if ent["cm"] is not None:
            ent["cm"].copy_(cm)
        ent["idx"].zero_()
        if seed is not None:
            ent["rng"].state.copy_(torch.tensor([seed & (2**63 - 1), 0], dtype=torch.int64))
        for _ in range(ent["n"]):
            ent["graph"].replay()
        out = [x.clone() for x in ent["x"]]
        return smp._postprocess(out if nested else out[0], clip=True)

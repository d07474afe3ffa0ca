"""Generate tests/golden/*.pt from the REAL reference (apple/ml-mdm) -- build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

For every case of tests/parity_cases.py: load the case's parameters into the reference's own
``UNet`` / ``NestedUNet`` (through ``state_dict`` -- which also proves key/shape compatibility),
run the reference forward + backward on CPU fp32 and store
  outputs          full tensors
  grad_norm[k]     ||dL/dp_k||            for every parameter
  grad_probe[k]    <dL/dp_k, probe_k>     (probe regenerated from the key at test time)
  grad_full[k]     full gradient for a few small tensors
  param_sum[k]     float64 sum of every parameter (regeneration check)
TEST INFRASTRUCTURE: not imported by the product.
"""
import dataclasses
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("ml-mdm_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
sys.dont_write_bytecode = True

import parity_cases as PC  # noqa: E402
import ref_import  # noqa: E402


def to_ref_cfg(R, cfg):
    d = {f.name: getattr(cfg, f.name) for f in dataclasses.fields(cfg)}
    rc = R.unet.ResNetConfig(**dataclasses.asdict(d.pop("resnet_config")))
    inner = d.pop("inner_config", None)
    if inner is None:
        return R.unet.UNetConfig(resnet_config=rc, **d)
    icfg = to_ref_cfg(R, inner)
    cls = R.nested_unet.Nested2UNetConfig if hasattr(icfg, "inner_config") else R.nested_unet.NestedUNetConfig
    return cls(resnet_config=rc, inner_config=icfg, **d)


def main():
    R = ref_import.load()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    only = [a for a in sys.argv[2:] if a in PC.ALL_CASES]
    for name in (only or PC.ALL_CASES):
        _, cfg, sd = PC.build_module(name)
        rcfg = to_ref_cfg(R, cfg)
        cls = R.nested_unet.NestedUNet if hasattr(rcfg, "inner_config") else R.unet.UNet
        ref = cls(3, 3, rcfg)
        missing, unexpected = ref.load_state_dict(sd, strict=True)
        assert not missing and not unexpected
        inp = PC.inputs(name)
        outs = ref(inp["x"], inp["times"], inp["cond"], inp["mask"], inp["micros"])
        PC.loss_of(outs, inp["gys"]).backward()
        grads = {k: p.grad for k, p in ref.named_parameters()}
        assert all(g is not None for g in grads.values())
        small = [k for k, g in grads.items() if g.numel() <= 256][:24]
        blob = {
            "case": name,
            "torch": torch.__version__,
            "outputs": [o.detach().clone() for o in PC.as_list(outs)],
            "grad_norm": {k: float(g.double().norm()) for k, g in grads.items()},
            "grad_probe": {k: float((g.double() * PC.probe_for(k, g.shape)).sum()) for k, g in grads.items()},
            "grad_full": {k: grads[k].clone() for k in small},
            "param_sum": {k: float(v.double().sum()) for k, v in sd.items()},
        }
        path = os.path.join(out_dir, name + ".pt")
        torch.save(blob, path)
        print("wrote", path, os.path.getsize(path), "bytes;", len(grads), "params; out mean|.| =",
              [float(o.abs().mean()) for o in blob["outputs"]])


def diffusion_host_golden():
    """outputs of the reference's Sampler / Diffusion / NestedDiffusion driven by tests/stub_models.py"""
    import numpy as np

    import stub_models as SM

    R = ref_import.load()
    S, D = R.samplers, R.diffusion
    blob = {"schedules": {}}
    for st in ("COSINE", "DDPM", "DEEPFLOYD"):
        smp = S.Sampler(S.SamplerConfig(num_diffusion_steps=1000, schedule_type=S.ScheduleType[st]))
        blob["schedules"][st] = {"gammas": smp.gammas.clone(), "vdm": smp.vdm_loss_weights.clone()}
    blob["timesteps_250"] = smp.set_timesteps(250)

    def sc(**kw):
        base = dict(num_diffusion_steps=1000, schedule_type=S.ScheduleType.DEEPFLOYD,
                    prediction_type=S.PredictionType.V_PREDICTION, loss_target_type=S.PredictionType.DDPM,
                    threshold_function=S.ThresholdType.CLIP)
        base.update(kw)
        return S.SamplerConfig(**base)

    g = torch.Generator().manual_seed(7)
    sample = {"images": torch.rand(3, 3, 16, 16, generator=g) * 2 - 1, "lm_outputs": torch.randn(3, 5, 8, generator=g),
              "lm_mask": torch.ones(3, 5)}
    # plain pipeline
    pipe = D.Diffusion(SM.StubUNet(), D.DiffusionConfig(sampler_config=sc(), use_vdm_loss_weights=False))
    torch.manual_seed(11)
    loss, time, x_t, means, tgt, w = pipe.get_loss(sample)
    blob["loss"] = dict(loss=loss.detach(), time=time, x_t=x_t.detach(), means=means.detach(), tgt=tgt.detach())
    pipe_v = D.Diffusion(SM.StubUNet(), D.DiffusionConfig(sampler_config=sc(), use_vdm_loss_weights=True))
    torch.manual_seed(11)
    blob["loss_vdm_weights"] = pipe_v.get_loss(sample)[5].detach()
    # round 3 (ADVICE): rescale_signal = 2 with a V-prediction TARGET -- x_t is built from the rescaled images, the
    # target from the raw ones (diffusion.py:153, 163)
    pipe_r = D.Diffusion(SM.StubUNet(), D.DiffusionConfig(
        sampler_config=sc(rescale_signal=2, loss_target_type=S.PredictionType.V_PREDICTION), use_vdm_loss_weights=False))
    torch.manual_seed(11)
    loss, time, x_t, means, tgt, w = pipe_r.get_loss(sample)
    blob["loss_rescale2_vtarget"] = dict(loss=loss.detach(), x_t=x_t.detach(), tgt=tgt.detach())
    for tag, kw in (("ddim", dict(ddim_eta=0)), ("ddpm", dict()), ("ddim_cfg", dict(ddim_eta=0, guidance_scale=3.0)),
                    ("ddpm_eta1_dyn", dict(ddim_eta=1))):
        cfg = sc(threshold_function=S.ThresholdType.DYNAMIC) if tag.endswith("dyn") else sc()
        pp = D.Diffusion(SM.StubUNet(), D.DiffusionConfig(sampler_config=cfg, use_vdm_loss_weights=False))
        smp_in = dict(sample)
        if "cfg" in tag:
            smp_in["lm_outputs"] = torch.cat([torch.zeros_like(sample["lm_outputs"]), sample["lm_outputs"]])
            smp_in["lm_mask"] = torch.cat([sample["lm_mask"]] * 2)
        torch.manual_seed(13)
        with torch.no_grad():
            out = pp.sample(3, smp_in, 16, torch.device("cpu"), resample_steps=True, num_inference_steps=4, **kw)
        blob["sample_" + tag] = out.detach()
    # nested pipeline (256-style: shifted schedule, double loss, no residual)
    ncfg = D.NestedDiffusionConfig(sampler_config=sc(schedule_shifted=True, rescale_signal=1), use_vdm_loss_weights=False,
                                   use_double_loss=True, no_use_residual=True, multi_res_weights="4:1")
    npipe = D.NestedDiffusion(SM.StubNestedUNet(), ncfg)
    nsample = dict(sample, images=torch.rand(3, 3, 32, 32, generator=g) * 2 - 1)
    torch.manual_seed(17)
    loss, time, x_t, pred, tgt, w = npipe.get_loss(nsample)
    blob["nested_loss"] = dict(loss=loss.detach(), time=time, x_t=x_t.detach(), pred=pred.detach(), tgt=tgt.detach())
    # round 3: mixed_ratio '2:1' (cc12m_256x256.yaml:108; diffusion.py:258-275, 374-381)
    mcfg = D.NestedDiffusionConfig(sampler_config=sc(schedule_shifted=True, rescale_signal=1), use_vdm_loss_weights=False,
                                   use_double_loss=True, no_use_residual=True, mixed_ratio="2:1")
    mpipe = D.NestedDiffusion(SM.StubNestedUNet(), mcfg)
    torch.manual_seed(17)
    loss, time, x_t, pred, tgt, w = mpipe.get_loss(nsample)
    blob["nested_loss_mixed"] = dict(loss=loss.detach(), x_t=x_t.detach(), pred=pred.detach(), tgt=tgt.detach())
    torch.manual_seed(19)
    with torch.no_grad():
        out = npipe.sample(3, nsample, 32, torch.device("cpu"), resample_steps=True, num_inference_steps=4, ddim_eta=0)
    blob["nested_sample_ddim"] = out.detach()
    torch.manual_seed(19)
    with torch.no_grad():
        out = npipe.sample(3, nsample, 32, torch.device("cpu"), resample_steps=True, num_inference_steps=3, output_inner=True)
    blob["nested_sample_ddpm_inner"] = out.detach()
    path = os.path.join(ROOT, "tests", "golden", "diffusion_host.pt")
    torch.save(blob, path)
    print("wrote", path, os.path.getsize(path), "bytes")


def sampling_golden():
    """4-step deterministic (DDIM eta=0) sampling and a train-step loss through the REAL reference pipeline +
    reference UNet / NestedUNet on the mini cases (BASELINE.json configs[0] shape of test, reduced size)."""
    R = ref_import.load()
    S, D = R.samplers, R.diffusion
    blob = {}
    for name in ("mini_unet", "mini_nested"):
        _, cfg, sd = PC.build_module(name)
        rcfg = to_ref_cfg(R, cfg)
        nested = hasattr(rcfg, "inner_config")
        ref = (R.nested_unet.NestedUNet if nested else R.unet.UNet)(3, 3, rcfg)
        ref.load_state_dict(sd, strict=True)
        scfg = S.SamplerConfig(num_diffusion_steps=1000, schedule_type=S.ScheduleType.DEEPFLOYD,
                               prediction_type=S.PredictionType.V_PREDICTION, loss_target_type=S.PredictionType.DDPM,
                               threshold_function=S.ThresholdType.CLIP, schedule_shifted=nested,
                               rescale_signal=1 if nested else None)
        if nested:
            pipe = D.NestedDiffusion(ref, D.NestedDiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False,
                                                                  use_double_loss=True, no_use_residual=True))
        else:
            pipe = D.Diffusion(ref, D.DiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False))
        inp = PC.inputs(name)
        side = 32 if nested else 16
        smp = {"lm_outputs": inp["cond"], "lm_mask": inp["mask"]}
        torch.manual_seed(23)
        with torch.no_grad():
            img = pipe.sample(2, smp, side, torch.device("cpu"), resample_steps=True, num_inference_steps=4, ddim_eta=0)
        g = torch.Generator().manual_seed(29)
        smp["images"] = torch.rand(2, 3, side, side, generator=g) * 2 - 1
        torch.manual_seed(31)
        pipe.train()
        loss = pipe.get_loss(smp)[0]
        blob[name] = {"sample": img.detach().clone(), "loss": loss.detach().clone()}
        print(name, "sample", tuple(img.shape), "loss", loss.tolist())
        if nested:
            # round 3: the shipped yaml key ``mixed_ratio: '2:1'`` (cc12m_256x256.yaml:108): the 32x32 level sees the first
            # int(2/3 * B) samples only, its loss is divided by 2/3 and zeroed for the rest (diffusion.py:258-275, 374-381);
            # B = 3 -> bh = 2, bl = 3.  Explicit micro-conditioning rides along (diffusion.py:136-141).
            mp = D.NestedDiffusion(ref, D.NestedDiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False,
                                                                use_double_loss=True, no_use_residual=True, mixed_ratio="2:1"))
            minp = PC.inputs("mini_nested_mixed")
            g = torch.Generator().manual_seed(37)
            msmp = {"lm_outputs": minp["cond"], "lm_mask": minp["mask"], "scale": minp["micros"]["scale"],
                    "images": torch.rand(3, 3, side, side, generator=g) * 2 - 1}
            torch.manual_seed(41)
            mp.train()
            mloss = mp.get_loss(msmp)[0]
            mloss.sum().backward()
            blob["mini_nested_mixed"] = {"loss": mloss.detach().clone(),
                                         "grad_norm": {k: float(p.grad.double().norm()) for k, p in ref.named_parameters()}}
            ref.zero_grad(set_to_none=True)
            print("mini_nested mixed_ratio 2:1 loss", mloss.tolist())
    path = os.path.join(ROOT, "tests", "golden", "pipeline.pt")
    torch.save(blob, path)
    print("wrote", path, os.path.getsize(path), "bytes")


def full_size_golden():
    """The three SHIPPED architectures at full size through the REAL reference (CPU fp32):
      unet64     (BASELINE.json configs[0]/[1]): forward B=2 + every parameter-gradient norm / probe, and the configs[0]
                 pipeline: Diffusion.sample() 4 DDIM steps at batch 2 + get_loss() on seeded inputs
      nested256  (configs[2]/[3]): forward + gradients, B=1
      nested1024 (configs[4]): forward B=1 (includes the x / std input normalisation of the 256 level)
    Outputs are kept as summaries (norm, seeded probe, strided subsample): the 1024^2 output alone is 12.6 MB."""
    import time

    R = ref_import.load()
    S, D = R.samplers, R.diffusion
    path = os.path.join(ROOT, "tests", "golden", "full_size.pt")
    only = [a for a in sys.argv[2:] if a in PC.FULL]
    blob = torch.load(path, weights_only=False) if (only and os.path.exists(path)) else {"torch": torch.__version__}
    for name in (only or PC.FULL):
        t0 = time.time()
        _, sd = PC.full_module(name)
        rcfg = to_ref_cfg(R, PC.full_cfg(name))
        nested = hasattr(rcfg, "inner_config")
        ref = (R.nested_unet.NestedUNet if nested else R.unet.UNet)(3, 3, rcfg)
        ref.load_state_dict(sd, strict=True)
        inp = PC.full_inputs(name)
        with_grad = True   # round 3: nested1024 too (all three levels' backward at full size, B = 1)
        ent = {"param_sum": {k: float(v.double().sum()) for k, v in list(sd.items())[::37]}}
        if with_grad:
            outs = PC.as_list(ref(inp["x"], inp["times"], inp["cond"], inp["mask"]))
            PC.loss_of(outs, inp["gys"]).backward()
            grads = {k: p.grad for k, p in ref.named_parameters()}
            ent["grad_norm"] = {k: float(g.double().norm()) for k, g in grads.items()}
            ent["grad_probe"] = {k: float((g.double() * PC.probe_for(k, g.shape)).sum()) for k, g in grads.items()}
            ref.zero_grad(set_to_none=True)
        else:
            with torch.no_grad():
                outs = PC.as_list(ref(inp["x"], inp["times"], inp["cond"], inp["mask"]))
        ent["outputs"] = [PC.summarize_output(o, "%s.out%d" % (name, i)) for i, o in enumerate(outs)]
        if name == "unet64":
            # BASELINE.json configs[0]: cc12m_64x64, batch 2, 4 diffusion steps, random T5 embeddings, CPU reference
            scfg = S.SamplerConfig(num_diffusion_steps=1000, schedule_type=S.ScheduleType.DEEPFLOYD,
                                   prediction_type=S.PredictionType.V_PREDICTION, loss_target_type=S.PredictionType.DDPM,
                                   threshold_function=S.ThresholdType.CLIP)
            pipe = D.Diffusion(ref, D.DiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False))
            smp = {"lm_outputs": inp["cond"], "lm_mask": inp["mask"]}
            torch.manual_seed(23)
            with torch.no_grad():
                img = pipe.sample(2, smp, 64, torch.device("cpu"), resample_steps=True, num_inference_steps=4, ddim_eta=0)
            g = torch.Generator().manual_seed(29)
            smp["images"] = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
            torch.manual_seed(31)
            pipe.train()
            loss = pipe.get_loss(smp)[0]
            ent["sample"] = img.detach().clone()
            ent["loss"] = loss.detach().clone()
        blob[name] = ent
        print(name, "done in %.1f s; out norms" % (time.time() - t0), [o["norm"] for o in ent["outputs"]])
        del ref
    torch.save(blob, path)
    print("wrote", path, os.path.getsize(path), "bytes")


class RecordingLogger:
    def __init__(self):
        self.rows = []

    def add_scalar(self, name, value):
        self.rows.append((name, float(value)))


def trainer_schedule(n_micro=6, accumulations=2):
    """(accumulate_gradient flag per micro-step) as clis/train_parallel.py:183-186 derives it"""
    flags, counter = [], 0
    for _ in range(n_micro):
        counter = (counter + 1) % accumulations
        flags.append(counter != 0)
    return flags


def run_train_batch(trainer_mod, pipe, ema_cls, fp16=False, n_micro=6, accumulations=2, nan_at=None):
    """drive ``trainer_mod.train_batch`` the way clis/train_parallel.py:122-230 does (AdamW, warm-up LambdaLR, ModelEma,
    gradient accumulation); shared by the golden generator (reference modules) and the tests (mdm_hip modules)"""
    import types

    vm = pipe.model.vision_model
    opt = torch.optim.AdamW(vm.parameters(), lr=1e-2, weight_decay=0, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: min(1.0, (it + 1) / 3))
    ema = ema_cls(vm, decay=0.9, warmup_steps=1)
    logger = RecordingLogger()
    args = types.SimpleNamespace(fp16=fp16, gradient_clip_norm=0.05)
    g = torch.Generator().manual_seed(7)
    sample = {"images": torch.rand(3, 3, 16, 16, generator=g) * 2 - 1, "lm_outputs": torch.randn(3, 5, 8, generator=g),
              "lm_mask": torch.ones(3, 5)}
    dev = next(vm.parameters()).device
    sample = {k: v.to(dev) for k, v in sample.items()}
    vals = []
    for i, acc in enumerate(trainer_schedule(n_micro, accumulations)):
        torch.manual_seed(100 + i)
        smp = dict(sample)
        if nan_at == i:
            smp["images"] = sample["images"] * float("nan")
        out = trainer_mod.train_batch(pipe, smp, opt, sched, logger, args, grad_scaler=None, accumulate_gradient=acc,
                                      num_grad_accumulations=accumulations, ema_model=ema, loss_factor=1.0)
        vals.append(out[0])
        assert len(out) == 6
    return {"loss_vals": vals, "w": float(vm.w), "ema_w": float(ema.module.w), "log": logger.rows, "lr": sched.get_last_lr()[0],
            "exp_avg": float(opt.state[vm.w]["exp_avg"]), "exp_avg_sq": float(opt.state[vm.w]["exp_avg_sq"]),
            "ema_counter": ema.counter}


def trainer_golden():
    """the REAL ``ml_mdm.trainer.train_batch`` + ``ModelEma`` + ``Diffusion`` around tests/stub_models.StubUNet, fp32 on CPU:
    6 micro-steps with 2-step gradient accumulation (3 optimizer steps), and the same with a NaN batch at micro-step 2"""
    import stub_models as SM

    R = ref_import.load()
    S, D = R.samplers, R.diffusion
    scfg = S.SamplerConfig(num_diffusion_steps=1000, schedule_type=S.ScheduleType.DEEPFLOYD,
                           prediction_type=S.PredictionType.V_PREDICTION, loss_target_type=S.PredictionType.DDPM)
    blob = {}
    for tag, nan_at in (("plain", None), ("nan", 2)):
        pipe = D.Diffusion(SM.StubUNet(), D.DiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False))
        blob[tag] = run_train_batch(R.trainer, pipe, R.model_ema.ModelEma, nan_at=nan_at)
        print(tag, blob[tag])
    path = os.path.join(ROOT, "tests", "golden", "train_batch.pt")
    torch.save(blob, path)
    print("wrote", path)


class _PhiloxFeed:
    """stands in for torch.randn_like inside the REFERENCE sampler: hands out, draw by draw, exactly the numbers the HIP
    step kernel generates from DeviceRng(seed) (element i = lane i & 3 of Philox block offset + i // 4, stream 0; the
    offset advances by ceil(n / 4) per draw: ml-mdm_amd/mdm_hip/samplers.py:_xt_last_hip) -- oracle/philox_ref.py"""

    def __init__(self, seed):
        import philox_ref as P

        self.P, self.seed, self.offset, self.draws = P, seed, 0, 0

    def __call__(self, like, **kw):
        n = like.numel()
        v = torch.from_numpy(self.P.normals(n, self.seed, self.offset, 0)).reshape(like.shape).to(like.dtype)
        self.offset += (n + 3) // 4
        self.draws += 1
        return v


LONG_CASES = {
    # name: (architecture, batch, steps, ddim_eta, sampler kwargs)
    # BASELINE.json configs[4]: flickr1024 NestedUNet (64+256+1024) DDPM sampling (generate_sample.py:546-551 with
    # ddim_eta=1 == ancestral DDPM, samplers.py:300) from a checkpoint round-tripped through UNet.save / UNet.load
    # (the real vis_model_1024x1024.pth is not on disk and there is no network: synthetic weights)
    "nested1024_ddpm": ("nested1024", 1, 25, 1, dict(schedule_shifted=True, schedule_shifted_power=2, rescale_signal=1)),
    # configs[4] as stated: the demo's full 250 DDPM steps (generate_sample.py:546-551), B = 1 (round 6; ~40 min of CPU once)
    "nested1024_ddpm250": ("nested1024", 1, 250, 1, dict(schedule_shifted=True, schedule_shifted_power=2, rescale_signal=1)),
    # long horizons on the 64x64 U-Net: the demo's 50 steps, ancestral DDPM (ddim_eta=None) and DDIM eta=0
    "unet64_ddpm50": ("unet64", 1, 50, None, {}),
    "unet64_ddim100": ("unet64", 1, 100, 0, {}),
}
LONG_SEED = 20240925


def long_start_noise(name):
    arch, B, _, _, _ = LONG_CASES[name]
    g = torch.Generator().manual_seed(77)
    return [torch.randn(B, 3, sd, sd, generator=g) for sd in PC.FULL[arch][1]]


def long_sampling_golden():
    """Long-horizon sampling through the REAL reference pipeline (CPU fp32), the sampling noise injected from the host
    replay of the device generator so that the HIP sampler with DeviceRng(LONG_SEED) sees the very same numbers.
    Written to tests/golden/long_sampling.pt as output summaries (parity_cases.summarize_output)."""
    import tempfile
    import time

    R = ref_import.load()
    S, D = R.samplers, R.diffusion
    path = os.path.join(ROOT, "tests", "golden", "long_sampling.pt")
    blob = torch.load(path, weights_only=False) if os.path.exists(path) else {}
    only = [a for a in sys.argv[2:] if a in LONG_CASES]
    for name in (only or LONG_CASES):
        arch, B, steps, eta, skw = LONG_CASES[name]
        t0 = time.time()
        ours, sd = PC.full_module(arch)
        rcfg = to_ref_cfg(R, PC.full_cfg(arch))
        nested = hasattr(rcfg, "inner_config")
        ref = (R.nested_unet.NestedUNet if nested else R.unet.UNet)(3, 3, rcfg)
        with tempfile.TemporaryDirectory() as td:
            # checkpoint written by OUR UNet.save, read by the REFERENCE's UNet.load (unet.py:794-832)
            ck = os.path.join(td, "vis_model_synthetic.pth")
            ours.save(ck, other_items={"batch_num": 7})
            items = ref.load(ck)
            assert items["batch_num"] == 7
        for k, v in ref.state_dict().items():
            assert torch.equal(v, sd[k]), k
        del ours
        scfg = S.SamplerConfig(num_diffusion_steps=1000, schedule_type=S.ScheduleType.DEEPFLOYD,
                               prediction_type=S.PredictionType.V_PREDICTION, loss_target_type=S.PredictionType.DDPM,
                               threshold_function=S.ThresholdType.CLIP, **skw)
        if nested:
            pipe = D.NestedDiffusion(ref, D.NestedDiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False,
                                                                  use_double_loss=True, no_use_residual=True))
        else:
            pipe = D.Diffusion(ref, D.DiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False))
        pipe.eval()
        inp = PC.full_inputs(arch)
        cond, mask = inp["cond"][:B], inp["mask"][:B]
        start = long_start_noise(name)
        feed = _PhiloxFeed(LONG_SEED)
        real, real_normal = torch.randn_like, torch.Tensor.normal_
        by_side = {t.shape[-1]: t for t in start[1:]}
        torch.randn_like = feed
        # the reference draws the start noise of the lower resolutions itself, with x_low.normal_() on its first step
        # (samplers.py:669-676): hand it the case's seeded start pyramid instead
        torch.Tensor.normal_ = lambda self, *a, **k: self.copy_(by_side[self.shape[-1]])
        try:
            with torch.no_grad():
                out = pipe.sampler.sample(pipe.get_model(), start[0].clone(), cond, mask, {},
                                          resample_steps=True, num_inference_steps=steps, ddim_eta=eta)
        finally:
            torch.randn_like, torch.Tensor.normal_ = real, real_normal
        blob[name] = {"out": PC.summarize_output(out, "long." + name), "draws": feed.draws, "shape": tuple(out.shape)}
        print(name, "done in %.1f s; %d noise draws; out norm %.4f mean|.| %.4f" % (
            time.time() - t0, feed.draws, blob[name]["out"]["norm"], float(out.abs().mean())), flush=True)
        torch.save(blob, path)
        del ref, pipe
    print("wrote", path, os.path.getsize(path), "bytes")


def reference_bf16_error():
    """The bar the bf16 mode is held to: the REFERENCE's own error when it runs under
    torch.autocast(bfloat16) -- what its `fp16: 1` training does (trainer.py:29-30) -- against its fp32 run, per parity
    case: rel-L2 of every output, and the aggregate rel-L2 of all parameter gradients (through the golden probes for the
    full-size cases).  CPU autocast; written to tests/golden/reference_bf16_error.pt."""
    import unet_oracle as O

    R = ref_import.load()
    res = {}

    def build(cfg, sd):
        rcfg = to_ref_cfg(R, cfg)
        ref = (R.nested_unet.NestedUNet if hasattr(rcfg, "inner_config") else R.unet.UNet)(3, 3, rcfg)
        ref.load_state_dict(sd, strict=True)
        return ref

    only = [a for a in sys.argv[2:] if a in PC.CASES or a in PC.FULL]
    path = os.path.join(ROOT, "tests", "golden", "reference_bf16_error.pt")
    if only:
        res = torch.load(path)
    for name in PC.CASES:
        if only and name not in only:
            continue
        _, cfg, sd = PC.build_module(name)
        ref, inp = build(cfg, sd), PC.inputs(name)
        o32 = PC.as_list(ref(inp["x"], inp["times"], inp["cond"], inp["mask"]))
        PC.loss_of(o32, inp["gys"]).backward()
        g32 = {k: p.grad.clone() for k, p in ref.named_parameters()}
        ref.zero_grad(set_to_none=True)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            o16 = PC.as_list(ref(inp["x"], inp["times"], inp["cond"], inp["mask"]))
        PC.loss_of([o.float() for o in o16], inp["gys"]).backward()
        num = sum(float((p.grad.double() - g32[k].double()).pow(2).sum()) for k, p in ref.named_parameters())
        den = sum(float(g.double().pow(2).sum()) for g in g32.values())
        res[name] = {"fwd": [O.rel_l2(a.float(), b) for a, b in zip(o16, o32)], "grad_agg": (num / den) ** 0.5}
        print(name, res[name], flush=True)
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "full_size.pt"), weights_only=False)
    for name in PC.FULL:
        if only and name not in only:
            continue
        _, sd = PC.full_module(name)
        ref, inp = build(PC.full_cfg(name), sd), PC.full_inputs(name)
        with_grad = "grad_norm" in gold[name]
        with torch.no_grad():
            o32 = PC.as_list(ref(inp["x"], inp["times"], inp["cond"], inp["mask"]))
        with torch.set_grad_enabled(with_grad), torch.autocast("cpu", dtype=torch.bfloat16):
            o16 = PC.as_list(ref(inp["x"], inp["times"], inp["cond"], inp["mask"]))
        res[name] = {"fwd": [O.rel_l2(a.float(), b) for a, b in zip(o16, o32)]}
        if with_grad:
            PC.loss_of([o.float() for o in o16], inp["gys"]).backward()
            g = gold[name]
            num = sum((float((p.grad.double() * PC.probe_for(k, p.grad.shape)).sum()) - g["grad_probe"][k]) ** 2
                      for k, p in ref.named_parameters())
            res[name]["grad_agg"] = (num / sum(n * n for n in g["grad_norm"].values())) ** 0.5
        print(name, res[name], flush=True)
    torch.save(res, path)


if __name__ == "__main__":
    which = sys.argv[1:2] or ["mini", "host", "pipeline", "full", "bf16", "long", "trainer"]
    if "mini" in which:
        main()
    if "host" in which:
        diffusion_host_golden()
    if "pipeline" in which:
        sampling_golden()
    if "full" in which:
        full_size_golden()
    if "bf16" in which:
        reference_bf16_error()
    if "long" in which:
        long_sampling_golden()
    if "trainer" in which:
        trainer_golden()

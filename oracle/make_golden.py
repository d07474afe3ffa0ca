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
    blob = {"torch": torch.__version__}
    for name in PC.FULL:
        t0 = time.time()
        _, sd = PC.full_module(name)
        rcfg = to_ref_cfg(R, PC.full_cfg(name))
        nested = hasattr(rcfg, "inner_config")
        ref = (R.nested_unet.NestedUNet if nested else R.unet.UNet)(3, 3, rcfg)
        ref.load_state_dict(sd, strict=True)
        inp = PC.full_inputs(name)
        with_grad = name != "nested1024"
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
    path = os.path.join(ROOT, "tests", "golden", "full_size.pt")
    torch.save(blob, path)
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

    for name in PC.CASES:
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
    torch.save(res, os.path.join(ROOT, "tests", "golden", "reference_bf16_error.pt"))


if __name__ == "__main__":
    which = sys.argv[1:] or ["mini", "host", "pipeline", "full", "bf16"]
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

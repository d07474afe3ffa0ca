"""CPU: the diffusion / sampler call-surface mirrors (mdm_hip.diffusion, mdm_hip.samplers) reproduce the
reference's Sampler / Diffusion / NestedDiffusion, checked against golden outputs produced by the real
reference (oracle/make_golden.py: diffusion_host_golden) with the same stub vision model and CPU RNG seed."""
import os

import pytest
import torch

import stub_models as SM
import unet_oracle as O

GOLD = torch.load(os.path.join(os.path.dirname(__file__), "golden", "diffusion_host.pt"), weights_only=False)


def sc(**kw):
    from mdm_hip import samplers as S

    base = dict(num_diffusion_steps=1000, schedule_type="DEEPFLOYD", prediction_type="V_PREDICTION",
                loss_target_type="DDPM", threshold_function="CLIP")
    base.update(kw)
    return S.SamplerConfig(**base)


def batch():
    g = torch.Generator().manual_seed(7)
    s = {"images": torch.rand(3, 3, 16, 16, generator=g) * 2 - 1, "lm_outputs": torch.randn(3, 5, 8, generator=g),
         "lm_mask": torch.ones(3, 5)}
    return s, g


@pytest.mark.parametrize("st", ["COSINE", "DDPM", "DEEPFLOYD"])
def test_noise_schedules(st):
    from mdm_hip import samplers as S

    smp = S.Sampler(S.SamplerConfig(num_diffusion_steps=1000, schedule_type=st))
    assert torch.equal(smp.gammas, GOLD["schedules"][st]["gammas"])
    assert torch.allclose(smp.vdm_loss_weights, GOLD["schedules"][st]["vdm"], rtol=1e-6, atol=0)
    assert (smp.set_timesteps(250) == GOLD["timesteps_250"]).all()


def test_get_loss_matches_reference():
    from mdm_hip import diffusion as D

    sample, _ = batch()
    pipe = D.Diffusion(SM.StubUNet(), D.DiffusionConfig(sampler_config=sc(), use_vdm_loss_weights=False))
    torch.manual_seed(11)
    loss, time, x_t, means, tgt, w = pipe.get_loss(sample)
    g = GOLD["loss"]
    assert w is None and torch.equal(time, g["time"])
    for a, b in ((loss, g["loss"]), (x_t, g["x_t"]), (means, g["means"]), (tgt, g["tgt"])):
        assert O.rel_l2(a, b) < 1e-6
    pipe_v = D.Diffusion(SM.StubUNet(), D.DiffusionConfig(sampler_config=sc(), use_vdm_loss_weights=True))
    torch.manual_seed(11)
    assert torch.allclose(pipe_v.get_loss(sample)[5], GOLD["loss_vdm_weights"])


def test_get_loss_rescaled_signal_with_v_target_matches_reference():
    """rescale_signal = 2 and a V-prediction loss target: x_t from the rescaled images, target from the raw ones
    (reference diffusion.py:153, 163)"""
    from mdm_hip import diffusion as D

    sample, _ = batch()
    pipe = D.Diffusion(SM.StubUNet(), D.DiffusionConfig(sampler_config=sc(rescale_signal=2, loss_target_type="V_PREDICTION"),
                                                        use_vdm_loss_weights=False))
    torch.manual_seed(11)
    loss, time, x_t, means, tgt, w = pipe.get_loss(sample)
    g = GOLD["loss_rescale2_vtarget"]
    for a, b in ((loss, g["loss"]), (x_t, g["x_t"]), (tgt, g["tgt"])):
        assert O.rel_l2(a, b) < 1e-6


def test_nested_mixed_ratio_loss_matches_reference():
    """mixed_ratio '2:1' (cc12m_256x256.yaml:108): the high resolution runs on int(2/3 B) samples, its loss is divided
    by 2/3 and zeroed for the rest (reference diffusion.py:258-275, 374-381)"""
    from mdm_hip import diffusion as D

    sample, g = batch()
    mcfg = D.NestedDiffusionConfig(sampler_config=sc(schedule_shifted=True, rescale_signal=1), use_vdm_loss_weights=False,
                                   use_double_loss=True, no_use_residual=True, mixed_ratio="2:1")
    pipe = D.NestedDiffusion(SM.StubNestedUNet(), mcfg)
    nsample = dict(sample, images=torch.rand(3, 3, 32, 32, generator=g) * 2 - 1)
    torch.manual_seed(17)
    loss, time, x_t, pred, tgt, w = pipe.get_loss(nsample)
    gl = GOLD["nested_loss_mixed"]
    for a, b in ((loss, gl["loss"]), (x_t, gl["x_t"]), (pred, gl["pred"]), (tgt, gl["tgt"])):
        assert O.rel_l2(a, b) < 1e-6
    assert float(loss[2]) > 0   # sample 2 keeps its low-resolution loss only


@pytest.mark.parametrize("tag,kw", [("ddim", dict(ddim_eta=0)), ("ddpm", dict()),
                                    ("ddim_cfg", dict(ddim_eta=0, guidance_scale=3.0)),
                                    ("ddpm_eta1_dyn", dict(ddim_eta=1))])
def test_sampling_loop_matches_reference(tag, kw):
    from mdm_hip import diffusion as D

    sample, _ = batch()
    cfg = sc(threshold_function="DYNAMIC") if tag.endswith("dyn") else sc()
    pipe = D.Diffusion(SM.StubUNet(), D.DiffusionConfig(sampler_config=cfg, use_vdm_loss_weights=False))
    if "cfg" in tag:
        sample["lm_outputs"] = torch.cat([torch.zeros_like(sample["lm_outputs"]), sample["lm_outputs"]])
        sample["lm_mask"] = torch.cat([sample["lm_mask"]] * 2)
    torch.manual_seed(13)
    with torch.no_grad():
        out = pipe.sample(3, sample, 16, torch.device("cpu"), resample_steps=True, num_inference_steps=4, **kw)
    assert O.rel_l2(out, GOLD["sample_" + tag]) < 1e-5


def test_nested_pipeline_matches_reference():
    from mdm_hip import diffusion as D

    sample, g = batch()
    ncfg = D.NestedDiffusionConfig(sampler_config=sc(schedule_shifted=True, rescale_signal=1), use_vdm_loss_weights=False,
                                   use_double_loss=True, no_use_residual=True, multi_res_weights="4:1")
    pipe = D.NestedDiffusion(SM.StubNestedUNet(), ncfg)
    nsample = dict(sample, images=torch.rand(3, 3, 32, 32, generator=g) * 2 - 1)
    torch.manual_seed(17)
    loss, time, x_t, pred, tgt, w = pipe.get_loss(nsample)
    gl = GOLD["nested_loss"]
    assert torch.equal(time, gl["time"])
    for a, b in ((loss, gl["loss"]), (x_t, gl["x_t"]), (pred, gl["pred"]), (tgt, gl["tgt"])):
        assert O.rel_l2(a, b) < 1e-6
    torch.manual_seed(19)
    with torch.no_grad():
        out = pipe.sample(3, nsample, 32, torch.device("cpu"), resample_steps=True, num_inference_steps=4, ddim_eta=0)
    assert O.rel_l2(out, GOLD["nested_sample_ddim"]) < 1e-5
    torch.manual_seed(19)
    with torch.no_grad():
        out = pipe.sample(3, nsample, 32, torch.device("cpu"), resample_steps=True, num_inference_steps=3, output_inner=True)
    assert out.shape == GOLD["nested_sample_ddpm_inner"].shape
    assert O.rel_l2(out, GOLD["nested_sample_ddpm_inner"]) < 1e-5


def test_train_step_runs_on_cpu_stub():
    """TrainStep glue (loss -> backward -> clip -> AdamW -> EMA) with the stub model, fp32 on CPU"""
    from mdm_hip import diffusion as D
    from mdm_hip.trainer import TrainStep

    sample, _ = batch()
    pipe = D.Diffusion(SM.StubUNet(), D.DiffusionConfig(sampler_config=sc(), use_vdm_loss_weights=False))
    step = TrainStep(pipe, bf16=False, lr=1e-2)
    w0 = float(pipe.get_model().vision_model.w)
    l0 = step(sample, time=torch.tensor(500))
    for _ in range(20):
        l1 = step(sample, time=torch.tensor(500))
    assert l1 < l0 and float(pipe.get_model().vision_model.w) != w0
    assert abs(float(step.ema[0]) - w0) < abs(float(pipe.get_model().vision_model.w) - w0)


@pytest.mark.reference
def test_registry_install_replaces_reference_models():
    import ref_import

    R = ref_import.load()
    import mdm_hip
    from mdm_hip import registry

    prev = registry.install(R.config)
    try:
        assert R.config.get_model("unet") is mdm_hip.UNet
        assert R.config.get_model("nested2_unet") is mdm_hip.NestedUNet
        # the reference's own config dataclass constructs our model (tests/test_models.py:33-44 of the reference)
        for name in ("unet", "nested_unet"):
            cfg_cls = R.config.MODEL_CONFIG_REGISTRY[name]["config"]
            m = R.config.get_model(name)(input_channels=3, output_channels=3, config=cfg_cls())
            assert sum(p.numel() for p in m.parameters()) > 0
        # and the reference pipeline wraps it
        pipe = R.config.get_pipeline("unet")(mdm_hip.UNet(3, 3, R.unet.UNetConfig()), R.diffusion.DiffusionConfig())
        assert pipe.get_model().vision_model.model_type == "unet"
    finally:
        R.config.MODEL_REGISTRY.update(prev)


def _oracle_forward(self, x_t, times, conditioning=None, cond_mask=None, micros=None):
    """stand-in for the HIP forward of mdm_hip.UNet / NestedUNet in the conformance test below: the oracle's restatement of
    the same function on the module's own parameters (TEST ONLY -- the product's ops raise on CPU tensors)"""
    sd = dict(self.named_parameters())   # (not state_dict(): that detaches)
    return O.model_forward(sd, self.config, x_t, times, conditioning, cond_mask, micros or {})


@pytest.mark.reference
@pytest.mark.parametrize("case", ["mini_unet", "mini_nested"])
def test_reference_pipeline_drives_the_hip_modules_call_surface(case, tmp_path, monkeypatch):
    """The REAL ``ml_mdm.diffusion.Diffusion`` / ``NestedDiffusion`` (reference diffusion.py:91-98, 295-313) wrapped around
    this package's ``UNet`` / ``NestedUNet`` -- the objects clis/train_parallel.py:66-72 builds once the registry is swapped
    -- run end to end on CPU: ``get_loss`` (noising, micro-conditioning dict, list inputs of the nested model, loss
    weights), ``sample`` (the sampler's calls into the vision model: tensor vs list conventions, ``nest_ratio``,
    ``is_temporal``, ``conditions``), ``save`` / ``load`` through the pipeline's ``Model`` (diffusion.py:66-70).  Only the arithmetic INSIDE the module's
    forward is replaced -- by the oracle, in this test -- so every attribute, argument and return convention the reference
    pipeline relies on is the product's; the same pipeline around the reference's own module with the same weights must
    give the same numbers."""
    import make_golden as MG
    import mdm_hip
    import parity_cases as PC
    import ref_import

    R = ref_import.load()
    monkeypatch.setattr(mdm_hip.UNet, "forward", _oracle_forward)
    monkeypatch.setattr(mdm_hip.NestedUNet, "forward", _oracle_forward)
    ours, cfg, sd = PC.build_module(case)
    nested = hasattr(cfg, "inner_config")
    ref_cls = R.nested_unet.NestedUNet if nested else R.unet.UNet
    theirs = ref_cls(3, 3, MG.to_ref_cfg(R, cfg))
    theirs.load_state_dict(sd)
    RS = R.samplers
    scfg = RS.SamplerConfig(num_diffusion_steps=1000, schedule_type=RS.ScheduleType.DEEPFLOYD,
                            prediction_type=RS.PredictionType.V_PREDICTION, loss_target_type=RS.PredictionType.DDPM,
                            threshold_function=RS.ThresholdType.CLIP)
    if nested:
        dcfg = R.diffusion.NestedDiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False, use_double_loss=True,
                                                 no_use_residual=True, multi_res_weights="4:1")
        P = R.diffusion.NestedDiffusion
    else:
        dcfg = R.diffusion.DiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False)
        P = R.diffusion.Diffusion
    pipe_o, pipe_r = P(ours, dcfg), P(theirs, dcfg)
    assert pipe_o.get_model().vision_model is ours and pipe_o.get_model().input_channels == 3

    side = PC._SIDES[case][0]
    g = torch.Generator().manual_seed(11)
    sample = {"images": torch.rand(2, 3, side, side, generator=g) * 2 - 1, "lm_outputs": torch.randn(2, 8, 64, generator=g),
              "lm_mask": torch.ones(2, 8), "scale": torch.tensor([0.3, 1.7])}
    sample["lm_mask"][1, 5:] = 0
    outs = []
    for pipe in (pipe_o, pipe_r):
        pipe.train()
        torch.manual_seed(23)
        outs.append(pipe.get_loss(sample))
    for a, b in zip(outs[0], outs[1]):
        if a is None or b is None:
            assert a is None and b is None
            continue
        for u, v in zip(a if isinstance(a, (list, tuple)) else [a], b if isinstance(b, (list, tuple)) else [b]):
            assert u.shape == v.shape and O.rel_l2(u.float(), v.float()) < 1e-4
    # the loss is differentiable w.r.t. the product module's parameters through the real pipeline
    outs[0][0].mean().backward()
    assert all(p.grad is not None for p in ours.parameters())

    # classifier-free guidance: the caller hands over [unconditional | conditional] text states (clis/generate_sample.py:230-256)
    cfg_sample = dict(sample, lm_outputs=torch.cat([torch.zeros_like(sample["lm_outputs"]), sample["lm_outputs"]]),
                      lm_mask=torch.cat([sample["lm_mask"]] * 2), scale=torch.cat([sample["scale"]] * 2))
    imgs = []
    for pipe in (pipe_o, pipe_r):
        pipe.eval()
        torch.manual_seed(29)
        with torch.no_grad():
            imgs.append(pipe.sample(2, cfg_sample, side, torch.device("cpu"), resample_steps=True, num_inference_steps=3,
                                    guidance_scale=1.5))
    assert imgs[0].shape == imgs[1].shape and O.rel_l2(imgs[0], imgs[1]) < 1e-4

    # checkpoint through the pipeline (diffusion.py:66-70): written by the product module, read by the reference's and back
    f = str(tmp_path / "vis.pth")
    pipe_o.get_model().save(f, other_items={"step": 3})
    theirs2 = ref_cls(3, 3, MG.to_ref_cfg(R, cfg))
    P(theirs2, dcfg).get_model().load(f)
    for (k, a), (_, b) in zip(sorted(ours.state_dict().items()), sorted(theirs2.state_dict().items())):
        assert torch.equal(a, b), k
    pipe_r.get_model().save(f)
    ours2, _, _ = PC.build_module(case, seed=5)
    P(ours2, dcfg).get_model().load(f)
    assert all(torch.equal(a, b) for a, b in zip(ours2.state_dict().values(), theirs.state_dict().values()))

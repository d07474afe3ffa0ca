#!/usr/bin/env python
"""Sampling latency of the three shipped architectures, eager sampler vs GraphedSampler (one hipGraph replay per whole
denoise iteration).  Development / reporting tool:
   gpurun -- python tools/sample_bench.py [unet64|nested256|nested1024] [batch] [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402

import mdm_hip  # noqa: E402
from mdm_hip import configs, diffusion, samplers  # noqa: E402
from mdm_hip.graph import GraphedSampler  # noqa: E402
from mdm_hip.testing import randomize_zero_params  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "nested1024"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n_it = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    dev = torch.device("cuda:0")
    sc = samplers.SamplerConfig(num_diffusion_steps=1000, schedule_type="DEEPFLOYD", prediction_type="V_PREDICTION",
                                loss_target_type="DDPM", schedule_shifted=which != "unet64", rescale_signal=1 if which != "unet64" else None,
                                schedule_shifted_power=2 if which == "nested1024" else 1)
    torch.manual_seed(0)
    if which == "unet64":
        net, side = mdm_hip.UNet(3, 3, configs.unet64_config(2048)), 64
        pipe = diffusion.Diffusion(net, diffusion.DiffusionConfig(sampler_config=sc, use_vdm_loss_weights=False))
    else:
        cfg = configs.nested256_config(2048) if which == "nested256" else configs.nested1024_config(2048)
        net, side = mdm_hip.NestedUNet(3, 3, cfg), 256 if which == "nested256" else 1024
        pipe = diffusion.NestedDiffusion(net, diffusion.NestedDiffusionConfig(sampler_config=sc, use_vdm_loss_weights=False,
                                                                              use_double_loss=True, no_use_residual=True))
    net.load_state_dict(randomize_zero_params(net.state_dict(), seed=1))
    pipe = pipe.to(dev)
    g = torch.Generator().manual_seed(1)
    smp = {"lm_outputs": torch.randn(batch, 32, 2048, generator=g).to(dev), "lm_mask": torch.ones(batch, 32).to(dev)}
    res = {"model": which, "batch": batch, "timed_steps": n_it, "sampler": "DDPM (ddim_eta=None), CFG off, bf16"}
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        pipe.sampler.use_device_rng(7, dev)
        pipe.sample(batch, smp, side, dev, resample_steps=True, num_inference_steps=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pipe.sample(batch, smp, side, dev, resample_steps=True, num_inference_steps=n_it)
        torch.cuda.synchronize()
        res["eager_ms_per_step"] = round((time.perf_counter() - t0) / n_it * 1e3, 3)
        gs = GraphedSampler(pipe, seed=7)
        gs.sample(batch, smp, side, dev, num_inference_steps=n_it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out2 = gs.sample(batch, smp, side, dev, num_inference_steps=n_it)
        torch.cuda.synchronize()
        res["graphed_ms_per_step"] = round((time.perf_counter() - t0) / n_it * 1e3, 3)
    res["finite"] = bool(torch.isfinite(out).all() and torch.isfinite(out2).all())
    res["max_mem_gb"] = round(torch.cuda.max_memory_allocated() / 2**30, 2)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Sampling latency of the three shipped architectures (development tool):
   gpurun -- python tools/sample_bench.py [unet64|nested256|nested1024] [batch] [graph]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402

import mdm_hip  # noqa: E402
from mdm_hip import configs, diffusion, samplers  # noqa: E402
from mdm_hip.testing import randomize_zero_params  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "nested1024"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    use_graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
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
    if use_graph:
        from mdm_hip.graph import GraphedDenoiser
        pipe.model.vision_model = GraphedDenoiser(pipe.model.vision_model)
    g = torch.Generator().manual_seed(1)
    smp = {"lm_outputs": torch.randn(batch, 32, 2048, generator=g).to(dev), "lm_mask": torch.ones(batch, 32).to(dev)}
    n_it = 8
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        pipe.sample(batch, smp, side, dev, resample_steps=True, num_inference_steps=3, ddim_eta=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pipe.sample(batch, smp, side, dev, resample_steps=True, num_inference_steps=n_it, ddim_eta=1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("%s B=%d graph=%s: %.2f ms / denoise step; 250-step image batch %.1f s; out %s finite=%s mem %.1f GB" % (
        which, batch, use_graph, dt / n_it * 1e3, dt / n_it * 250, tuple(out.shape), bool(torch.isfinite(out).all()),
        torch.cuda.max_memory_allocated() / 2**30))


if __name__ == "__main__":
    main()

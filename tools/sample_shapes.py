#!/usr/bin/env python
"""Per-shape kernel table of ONE denoise iteration of the nested 64+256+1024 sampler (configs[4], batch 4, bf16):
HIP events on the launch stream around every GEMM-class / HBM-class launch of the eager sampler.
   gpurun -- python tools/sample_shapes.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import mdm_hip  # noqa: E402
from mdm_hip import configs, diffusion, ops, samplers  # noqa: E402
from mdm_hip.testing import randomize_zero_params  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    IT = 4   # profiled denoise iterations; the table is per iteration
    dev = torch.device("cuda:0")
    sc = samplers.SamplerConfig(num_diffusion_steps=1000, schedule_type="DEEPFLOYD", prediction_type="V_PREDICTION",
                                loss_target_type="DDPM", schedule_shifted=True, rescale_signal=1, schedule_shifted_power=2)
    torch.manual_seed(0)
    net = mdm_hip.NestedUNet(3, 3, configs.nested1024_config(2048))
    net.load_state_dict(randomize_zero_params(net.state_dict(), seed=1))
    pipe = diffusion.NestedDiffusion(net, diffusion.NestedDiffusionConfig(sampler_config=sc, use_vdm_loss_weights=False,
                                                                          use_double_loss=True, no_use_residual=True)).to(dev)
    s = bench.synthetic_batch(B, 64, dev, seed=7)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        pipe.sample(B, s, 1024, dev, num_inference_steps=2, ddim_eta=1, resample_steps=True)
        torch.cuda.synchronize()
        ops.profile_begin(shapes=True)
        pipe.sample(B, s, 1024, dev, num_inference_steps=IT, ddim_eta=1, resample_steps=True)
        torch.cuda.synchronize()
        roof = ops.profile_end(bench.PEAK_BF16_TFLOPS)
    rows = sorted(roof["all_gemm_kernels"].items(), key=lambda kv: -kv[1]["time_ms"])
    tot = sum(v["time_ms"] for _, v in rows) / IT
    print("GEMM-class launches of one nested-1024 denoise iteration at batch %d: %.2f ms, %.1f TF/s FLOP-weighted" % (
        B, tot, roof["gemm_weighted"]["tflops"]))
    for k, v in rows[:40]:
        print("%8.3f ms  x%-3d %7.1f TF  %s" % (v["time_ms"] / IT, v["launches"] // IT, v["tflops"], k))
    print("HBM-class:")
    for k, v in roof["hbm_kernels"].items():
        print("%8.3f ms  x%-3d %7.0f GB/s  %s" % (v["time_ms"] / IT, v["launches"] // IT, v["gb_per_s"], k))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""ms per denoise iteration of the UNet-64 sampler at batch 64 on fp32 tensors with bf16x3 products (MDM_F32_SPLIT), eager:
   gpurun -- python tools/sample_x3_probe.py [iters]      (MDM_HIP_LIB selects a variant library)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from mdm_hip import ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    pipe, side = bench.build("unet64", dev)
    pipe.eval()
    g = torch.Generator().manual_seed(1)
    smp = {"lm_outputs": torch.randn(64, 32, 2048, generator=g).to(dev), "lm_mask": torch.ones(64, 32).to(dev)}
    with torch.no_grad(), ops.fp32_split(True):
        pipe.sample(64, smp, side, dev, resample_steps=True, num_inference_steps=1, ddim_eta=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.sample(64, smp, side, dev, resample_steps=True, num_inference_steps=iters, ddim_eta=0)
        torch.cuda.synchronize()
    print("x3 sampling lib=%s  %.2f ms per iteration" % (os.environ.get("MDM_HIP_LIB", "product"), (time.perf_counter() - t0) / iters * 1e3))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Time the REAL reference (apple/ml-mdm classes imported from /root/reference through oracle/ref_import.py) on this
machine's host cores -- BUILD CONTAINER ONLY (the reference tree does not exist on the GPU box).  Protocol of
SURVEY.md section 8d / BASELINE.md section 2: fp32, all host cores, 1 warm-up + 3 timed iterations, median;
train metric (Diffusion.get_loss + backward) at batch 2 and 8, sampling = configs[0] (UNet-64, batch 2, 4 steps).

    PYTHONDONTWRITEBYTECODE=1 python tools/cpu_reference_bench.py > profiles/r02_cpu_reference_real.json
"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("ml-mdm_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
sys.dont_write_bytecode = True
import torch  # noqa: E402

import make_golden as MG  # noqa: E402
import parity_cases as PC  # noqa: E402
import ref_import  # noqa: E402


def med(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def main():
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    R = ref_import.load()
    S, D = R.samplers, R.diffusion
    out = {"what": "apple/ml-mdm reference classes, CPU fp32, torch %s" % torch.__version__, "cores": cores,
           "protocol": "1 warm-up + 3 timed iterations, median", "results": {}}
    for name, nested in (("unet64", False), ("nested256", True)):
        _, sd = PC.full_module(name)
        ref = (R.nested_unet.NestedUNet if nested else R.unet.UNet)(3, 3, MG.to_ref_cfg(R, PC.full_cfg(name)))
        ref.load_state_dict(sd, strict=True)
        scfg = S.SamplerConfig(num_diffusion_steps=1000, schedule_type=S.ScheduleType.DEEPFLOYD,
                               prediction_type=S.PredictionType.V_PREDICTION, loss_target_type=S.PredictionType.DDPM,
                               threshold_function=S.ThresholdType.CLIP, schedule_shifted=nested, rescale_signal=1 if nested else None)
        if nested:
            pipe = D.NestedDiffusion(ref, D.NestedDiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False,
                                                                  use_double_loss=True, no_use_residual=True))
        else:
            pipe = D.Diffusion(ref, D.DiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False))
        side = 256 if nested else 64
        res = {}
        for b in ((2, 8) if not nested else (2,)):
            g = torch.Generator().manual_seed(1)
            smp = {"images": torch.rand(b, 3, side, side, generator=g) * 2 - 1, "lm_outputs": torch.randn(b, 32, 2048, generator=g),
                   "lm_mask": torch.ones(b, 32)}

            def train():
                pipe.train()
                ref.zero_grad(set_to_none=True)
                pipe.get_loss(smp)[0].mean().backward()

            t = med(train)
            res["train_fwd_bwd_batch%d" % b] = {"seconds": round(t, 3), "samples_per_s": round(b / t, 4)}
        g = torch.Generator().manual_seed(2)
        smp = {"lm_outputs": torch.randn(2, 32, 2048, generator=g), "lm_mask": torch.ones(2, 32)}

        def sample():
            with torch.no_grad():
                pipe.sample(2, smp, side, torch.device("cpu"), resample_steps=True, num_inference_steps=4, ddim_eta=0)

        t = med(sample)
        res["sample_4_steps_batch2"] = {"seconds": round(t, 3), "sample_steps_per_s": round(8 / t, 3)}
        out["results"][name] = res
        del ref, pipe
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""configs[4] (nested 64+256+1024 DDPM sampling, batch 4, bf16) for the HBM-traffic counters: one warm-up iteration (weight
packs included), then exactly N eager iterations.  Run under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate
passes) with two values of N; the difference of the totals divided by the difference of N is the traffic of ONE denoise
iteration (warm-up and packing cancel).  tools/sample_pmc.py sum <dir_f_N1> <dir_w_N1> <dir_f_N2> <dir_w_N2> N1 N2 out.json
folds the four passes into profiles/r06_pmc_nested1024_sampling.json (bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024, the
gfx950 correction of MI355X_MICROARCH.md); bench.py's nested1024_sampling leg reads it for its HBM fraction.
   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_f2 -o pmc -- python tools/sample_pmc.py run 2"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n_it):
    sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
    import torch

    import mdm_hip
    from mdm_hip import configs, diffusion, samplers
    from mdm_hip.testing import randomize_zero_params

    dev = torch.device("cuda:0")
    sc = samplers.SamplerConfig(num_diffusion_steps=1000, schedule_type="DEEPFLOYD", prediction_type="V_PREDICTION",
                                loss_target_type="DDPM", schedule_shifted=True, rescale_signal=1, schedule_shifted_power=2)
    torch.manual_seed(0)
    net = mdm_hip.NestedUNet(3, 3, configs.nested1024_config(2048))
    net.load_state_dict(randomize_zero_params(net.state_dict(), seed=1))
    pipe = diffusion.NestedDiffusion(net, diffusion.NestedDiffusionConfig(sampler_config=sc, use_vdm_loss_weights=False,
                                                                          use_double_loss=True, no_use_residual=True)).to(dev)
    g = torch.Generator().manual_seed(1)
    smp = {"lm_outputs": torch.randn(4, 32, 2048, generator=g).to(dev), "lm_mask": torch.ones(4, 32).to(dev)}
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        pipe.sampler.use_device_rng(7, dev)
        pipe.sample(4, smp, 1024, dev, resample_steps=True, num_inference_steps=1, ddim_eta=1)
        torch.cuda.synchronize()
        out = pipe.sample(4, smp, 1024, dev, resample_steps=True, num_inference_steps=n_it, ddim_eta=1)
        torch.cuda.synchronize()
    print("ran", n_it, "iterations, finite:", bool(torch.isfinite(out).all()))


def total(d, counter):
    t = 0.0
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                t += float(r["Counter_Value"])
    return t


def fold(f1, w1, f2, w2, n1, n2, out):
    b1 = (2 * total(f1, "FETCH_SIZE") + total(w1, "WRITE_SIZE")) * 1024
    b2 = (2 * total(f2, "FETCH_SIZE") + total(w2, "WRITE_SIZE")) * 1024
    per = (b2 - b1) / (n2 - n1)
    json.dump({"note": "nested 64+256+1024 DDPM sampling, batch 4, bf16, eager sampler: HBM bytes of ONE denoise iteration = (bytes of a run "
                       "with %d iterations - bytes of a run with %d) / %d; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE "
                       "counts half of a wide coalesced read, MI355X_MICROARCH.md), FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes "
                       "(tools/sample_pmc.py)" % (n2, n1, n2 - n1),
               "hbm_bytes_per_iteration": per, "bytes_run_n1": b1, "bytes_run_n2": b2, "n1": n1, "n2": n2}, open(out, "w"), indent=1)
    print("HBM bytes per denoise iteration: %.2f GB" % (per / 1e9))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        fold(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7]), sys.argv[8])

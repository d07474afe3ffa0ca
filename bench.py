#!/usr/bin/env python
"""Headline benchmark: denoise train steps/s of the cc12m_64x64 U-Net (bf16, batch 64 per GPU).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic batch: Diffusion.get_loss (noising +
U-Net forward) -> backward -> gradient all-reduce (N > 1) -> clip -> AdamW -> EMA, i.e. the
reference's ``trainer.train_batch`` (trainer.py:13-96).  Inputs are resident in HBM before the
timed region.  Rank 0 prints ONE JSON line (contract in the task statement) with two extra
objects: ``roofline`` (dominant kernel, measured live with HIP events on the launch stream)
and ``cpu_baseline`` (the CPU oracle timed on this box's host cores, rank 0 / N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "ml-mdm_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic forward matmul-class FLOPs per sample, measured on the reference (SURVEY.md section 8a/8d, S = 32)
FWD_GFLOP_PER_SAMPLE = {"unet64": 363.9, "nested256": 589.1, "mini": 0.0}
PEAK_BF16_TFLOPS = 2516.6   # 256 CU x 4096 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_F32_TFLOPS = 157.3


def build(workload, device, seed=0):
    import unet_oracle as O
    import mdm_hip
    from mdm_hip import configs, diffusion, samplers

    sc = samplers.SamplerConfig(num_diffusion_steps=1000, schedule_type="DEEPFLOYD", prediction_type="V_PREDICTION",
                                loss_target_type="DDPM", threshold_function="CLIP")
    torch.manual_seed(seed)
    if workload == "unet64":
        net = mdm_hip.UNet(3, 3, configs.unet64_config(2048))
        pipe_cls, dcfg, side = diffusion.Diffusion, diffusion.DiffusionConfig(sampler_config=sc, use_vdm_loss_weights=False), 64
    elif workload == "mini":  # development only (reduced architecture, not a reportable number)
        net = mdm_hip.UNet(3, 3, configs.mini_unet_config(2048))
        pipe_cls, dcfg, side = diffusion.Diffusion, diffusion.DiffusionConfig(sampler_config=sc, use_vdm_loss_weights=False), 16
    else:
        sc.schedule_shifted, sc.rescale_signal = True, 1
        net = mdm_hip.NestedUNet(3, 3, configs.nested256_config(2048))
        dcfg = diffusion.NestedDiffusionConfig(sampler_config=sc, use_vdm_loss_weights=False, use_double_loss=True, no_use_residual=True)
        pipe_cls, side = diffusion.NestedDiffusion, 256
    # the reference zero-initialises ~40% of the tensors; randomise them (seeded) so no work is degenerate
    net.load_state_dict(O.randomize_zero_params(net.state_dict(), seed=4321))
    pipe = pipe_cls(net, dcfg).to(device)
    return pipe, side


def synthetic_batch(batch, side, device, seed):
    g = torch.Generator().manual_seed(seed)
    return {
        "images": (torch.rand(batch, 3, side, side, generator=g) * 2 - 1).to(device),
        "lm_outputs": torch.randn(batch, 32, 2048, generator=g).to(device),
        "lm_mask": torch.ones(batch, 32).to(device),
    }


def pmc_traffic(kernel_label):
    """HBM bytes per launch of the named kernel from the committed PMC passes (profiles/r01_pmc_hbm_traffic.json, built
    by tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this script, with the gfx950
    FETCH_SIZE x2 correction).  ``kernel_label`` is the kernel name as rocprofv3 prints it (without arguments);
    None when there is no measurement for it."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
    if not os.path.exists(path):
        return None
    table = json.load(open(path))["kernels"]
    for k, v in table.items():
        if kernel_label in k:
            return v["hbm_bytes_per_launch"]
    return None


def cpu_baseline(workload, batch_ref):
    """the oracle (CPU restatement of the reference path) fwd+bwd on the host cores; bounded sample"""
    import unet_oracle as O
    import mdm_hip
    from mdm_hip import configs

    # torch's intra-op pool degrades badly beyond a few dozen threads on this op mix (632 s at 256 threads
    # vs ~10 s at 8 for the same work), so the baseline uses at most 32 host threads and reports that count
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    b = 8 if workload == "unet64" else 2   # ~10 s of host work at 32 threads
    side = 64 if workload == "unet64" else 256
    torch.manual_seed(0)
    cfg = configs.unet64_config(2048) if workload == "unet64" else configs.nested256_config(2048)
    cls = mdm_hip.UNet if workload == "unet64" else mdm_hip.NestedUNet
    sd = O.randomize_zero_params(cls(3, 3, cfg).state_dict(), seed=4321)
    leaf = {k: v.requires_grad_(True) for k, v in sd.items()}
    cfg = configs.unet64_config(2048) if workload == "unet64" else configs.nested256_config(2048)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, 3, side, side, generator=g)
    if workload != "unet64":
        x = [x, torch.randn(b, 3, 64, 64, generator=g)]
    cond, mask = torch.randn(b, 32, 2048, generator=g), torch.ones(b, 32)
    t0 = time.perf_counter()
    out = O.model_forward(leaf, cfg, x, torch.randint(0, 1000, (b,), generator=g), cond, mask)
    loss = sum(o.square().mean() for o in (out if isinstance(out, list) else [out]))
    loss.backward()
    dt = time.perf_counter() - t0
    return {
        "value": round((b / batch_ref) / dt, 6),
        "unit": "denoise-steps/s (batch %d equivalent)" % batch_ref,
        "cores": cores,
        "kind": "port",
        "sample": "1 un-warmed fwd+bwd of the fp32 CPU oracle at batch %d (%.1f s), scaled by %d/%d" % (b, dt, b, batch_ref),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="unet64", choices=["unet64", "nested256", "mini"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 64 / 16)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sampling", action="store_true")
    ap.add_argument("--sample-batch", type=int, default=None)
    args = ap.parse_args()

    from mdm_hip import distributed as mdist
    from mdm_hip import ops
    from mdm_hip.trainer import TrainStep

    # MDM_DIST_BACKEND=gloo + MDM_BENCH_DEVICE=0 let two ranks share ONE GPU (development check of the N > 1 path
    # on the single-GPU box); the driver's real runs use RCCL with one GPU per rank
    local, rank, world = mdist.init_distributed_singlenode(backend=os.environ.get("MDM_DIST_BACKEND"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if "MDM_BENCH_DEVICE" in os.environ:
        local = int(os.environ["MDM_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    batch = args.batch or {"unet64": 64, "nested256": 16, "mini": 4}[args.workload]

    pipe, side = build(args.workload, device)
    step = TrainStep(pipe, bf16=args.dtype == "bf16")
    sample = synthetic_batch(batch, side, device, seed=1234 + rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(sample)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(sample)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # secondary metric: sampling throughput (replicas only: every rank samples its own prompts, no collective)
    samp = None
    if not args.no_sampling:
        n_it = 6
        pipe.eval()
        sb = args.sample_batch or batch
        ssample = synthetic_batch(sb, side, device, seed=4321 + rank)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype == "bf16"):
            pipe.sample(sb, ssample, side, device, resample_steps=True, num_inference_steps=2, ddim_eta=0)  # warm-up
            sync()
            ts = time.perf_counter()
            pipe.sample(sb, ssample, side, device, resample_steps=True, num_inference_steps=n_it, ddim_eta=0)
            sync()
            dts = time.perf_counter() - ts
        ms_it = dts / n_it * 1e3
        demo_steps = 50 if args.workload == "unet64" else 100   # generate_sample.py:546-551 demo defaults
        samp = {"ms_per_denoise_step": round(ms_it, 3), "batch_per_gpu": sb, "images_per_s_at_%d_steps" % demo_steps:
                round(world * sb / (demo_steps * ms_it / 1e3), 3), "sampler": "DDIM eta=0, CFG off", "timed_steps": n_it}
    roof = None
    if not args.no_roofline:
        # one extra, untimed step with HIP events around every GEMM-class launch (on the launch stream).  Every rank
        # runs the step (it contains the gradient all-reduce); only rank 0 records and reports.
        pipe.train()
        if rank == 0:
            ops.profile_begin()
        step(sample)
        torch.cuda.synchronize()
        if rank == 0:
            roof = ops.profile_end(PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS)
            if roof is not None and args.workload == "unet64" and args.dtype == "bf16" and batch == 64:
                roof["traffic"] = pmc_traffic(roof["kernel"])   # HBM bytes per launch (PMC), null if not collected
    if world > 1:
        dist.barrier()

    if rank == 0:
        ms = dt / args.steps * 1e3
        steps_per_s = world * args.steps / dt
        alg_tflop_step = 3 * FWD_GFLOP_PER_SAMPLE[args.workload] * batch / 1e3
        out = {
            "metric": "denoise-steps/sec (train fwd+bwd+optimizer, %s, batch %d per GPU)" % (
                "cc12m_64x64 U-Net" if args.workload == "unet64" else "cc12m_256x256 NestedUNet", batch),
            "value": round(steps_per_s, 4),
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (U(-1,1) images, N(0,1) text states [B,32,2048], random-init weights with zero-init tensors randomised)",
            "config": {
                "workload": "%s train step, per-GPU batch %d, global batch %d" % (args.workload, batch, batch * world),
                "global_batch": batch * world,
                "parallelism": "dp%d" % world,
                "samples_per_s": round(steps_per_s * batch, 2),
                "step_algorithmic_tflop": round(alg_tflop_step, 2),
                "step_mfma_roofline_frac": round(alg_tflop_step / (ms / 1e3) / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS), 4),
            },
            "roofline": roof,
            "sampling": samp,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Headline benchmark: denoise train steps/s of the cc12m_64x64 U-Net (bf16, batch 64 per GPU).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic batch: Diffusion.get_loss (noising +
U-Net forward + loss) -> backward -> gradient all-reduce (N > 1) -> clip -> AdamW -> EMA, i.e. the
reference's ``trainer.train_batch`` (trainer.py:13-96).  Inputs are resident in HBM before the
timed region.  Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline      the GEMM-class kernel with the largest total time of a step, measured live with HIP events on the
                launch stream; ``gemm_weighted`` = FLOP-weighted figure over every GEMM-class launch;
                ``hbm_kernels`` = the streaming kernels as algorithmic GB/s against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (a port of the reference path) on this box's host cores, rank 0 / N = 1 only
  sampling      ms per denoise step (DDIM, whole iteration: model + fused update)
  nested256     BASELINE.json configs[2] (cc12m_256x256 NestedUNet, bf16, batch 16): steps/s of the same train step
  reference_loop  the SAME entry point on its plain path (autograd gradient accumulation, GradScaler, clip_grad_norm_,
                torch AdamW, per-tensor ModelEma -- what the unchanged CLI objects do without the arena adoption)
  nested1024_sampling  BASELINE.json configs[4]: ms per DDPM iteration of the 64+256+1024 NestedUNet at batch 4 and
                the 250-step total (generate_sample.py:546-551)

The timed step IS ``mdm_hip.trainer.train_batch`` -- the reference's ``trainer.train_batch`` signature
(trainer.py:13-25) called the way clis/train_parallel.py:122-230 calls it (torch AdamW lr 5e-5, warm-up LambdaLR,
ModelEma 0.9999, gradient_clip_norm 2, bf16 autocast): one import swap in the CLI gives this number.
"""
import argparse
import gc
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if os.path.join(ROOT, "ml-mdm_amd") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic forward matmul-class FLOPs per sample, measured on the reference (SURVEY.md section 8a/8d, S = 32)
FWD_GFLOP_PER_SAMPLE = {"unet64": 363.9, "nested256": 589.1, "mini": 0.0}
PEAK_BF16_TFLOPS = 2516.6   # 256 CU x 4096 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0       # HBM3E spec (MI355X_MICROARCH.md; 6.29 TB/s measured for a float4 copy)
PMC_FILES = ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json")


def build(workload, device, seed=0):
    import mdm_hip
    from mdm_hip import configs, diffusion, samplers
    from mdm_hip.testing import randomize_zero_params

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
    net.load_state_dict(randomize_zero_params(net.state_dict(), seed=4321))
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
    """(HBM bytes per launch, source file) of the named kernel from the committed PMC passes (profiles/rNN_pmc_hbm_traffic.json,
    built by tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this script, with the
    gfx950 FETCH_SIZE x2 correction).  PMC collection needs rocprofv3, so this is a lookup, not a live measurement."""
    for name in PMC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        # the live label ends its template list where the defaulted flags begin ("<256, 256, 2, 4, 0>" against the
        # symbol's "<256, 256, 2, 4, 0, false, false, false, false>"): match up to the closing bracket
        stem = kernel_label[:-1] if kernel_label.endswith(">") else kernel_label
        for k, v in json.load(open(path))["kernels"].items():
            if kernel_label in k or (stem + ", false" in k and "true" not in k[k.find(stem):]):
                return v["hbm_bytes_per_launch"], "profiles/" + name
    return None, None


def cpu_baseline(workload, batch_ref):
    """The CPU oracle (oracle/unet_oracle.py, a port of the reference path pinned to it by tests/golden) fwd+bwd on the
    host cores: per batch size 1 warm-up + timed iterations, median.  The real reference classes timed in the build
    container are in profiles/r02_cpu_reference_real.json (the reference tree does not exist on the GPU box)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import unet_oracle as O
    import mdm_hip
    from mdm_hip import configs

    # torch's intra-op pool degrades badly beyond a few dozen threads on this op mix (632 s at 256 threads
    # vs ~10 s at 8 for the same work), so the baseline uses at most 32 host threads and reports that count
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    side = 64 if workload == "unet64" else 256
    torch.manual_seed(0)
    cfg_fn = (lambda: configs.unet64_config(2048)) if workload == "unet64" else (lambda: configs.nested256_config(2048))
    cls = mdm_hip.UNet if workload == "unet64" else mdm_hip.NestedUNet
    sd = O.randomize_zero_params(cls(3, 3, cfg_fn()).state_dict(), seed=4321)
    leaf = {k: v.requires_grad_(True) for k, v in sd.items()}
    per_batch = {}
    for b, timed in ((2, 3), (8, 2)) if workload == "unet64" else ((2, 2),):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(b, 3, side, side, generator=g)
        if workload != "unet64":
            x = [x, torch.randn(b, 3, 64, 64, generator=g)]
        cond, mask, t = torch.randn(b, 32, 2048, generator=g), torch.ones(b, 32), torch.randint(0, 1000, (b,), generator=g)
        times = []
        for it in range(1 + timed):
            for v in leaf.values():
                v.grad = None
            t0 = time.perf_counter()
            out = O.model_forward(leaf, cfg_fn(), x, t, cond, mask)
            sum(o.square().mean() for o in (out if isinstance(out, list) else [out])).backward()
            if it > 0:
                times.append(time.perf_counter() - t0)
        per_batch[b] = statistics.median(times)
    b_big = max(per_batch)
    return {
        "value": round((b_big / batch_ref) / per_batch[b_big], 6),
        "unit": "denoise-steps/s (batch %d equivalent)" % batch_ref,
        "cores": cores,
        "kind": "port",
        "sample": "fp32 CPU oracle fwd+bwd, 1 warm-up + median of the timed iterations: " + ", ".join(
            "batch %d %.2f s (%.3f samples/s)" % (b, s, b / s) for b, s in sorted(per_batch.items())) +
            "; value = batch-%d rate scaled by %d/%d" % (b_big, b_big, batch_ref),
        "samples_per_s": {str(b): round(b / s, 4) for b, s in per_batch.items()},
        "real_reference": _real_reference(workload, batch_ref),
    }


def _real_reference(workload, batch_ref):
    """the REAL reference classes' figures next to the port's (the reference tree does not exist on the GPU box, so they were
    timed where it does -- the build container, 8 cores -- with tools/cpu_reference_bench.py and are carried as a record)"""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r02_cpu_reference_real.json")))
        res = rec["results"][workload]
        key = max((k for k in res if k.startswith("train_fwd_bwd_batch")), key=lambda k: int(k.rsplit("batch", 1)[1]))
        sps = float(res[key]["samples_per_s"])
        return {"kind": "reference", "what": rec["what"], "cores": rec["cores"], "where": "build container (not the GPU box)",
                "sample": "%s: %.2f s, %s" % (key, res[key]["seconds"], rec["protocol"]), "samples_per_s": sps,
                "value": round(sps / batch_ref, 6), "unit": "denoise-steps/s (batch %d equivalent)" % batch_ref,
                "file": "profiles/r02_cpu_reference_real.json"}
    except (OSError, KeyError, ValueError):
        return {"file": "profiles/r02_cpu_reference_real.json"}


def make_step(pipe, bf16, world, plain=False, bucket_mb=256.0, wire="auto", serial_wgrad=False, force_collectives=False, torch_ddp=False):
    """The objects clis/train_parallel.py:107-154 builds around the pipeline, and the step closure of its loop
    (:216-230): -> (step(sample) -> loss value, optimizer)"""
    import types

    from mdm_hip import distributed as mdist
    from mdm_hip import trainer

    vm = pipe.model.vision_model
    opt = torch.optim.AdamW(vm.parameters(), lr=5e-5, weight_decay=0, eps=1e-8)                 # :122-128
    warm = 1000
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: float(max(1, it)) / warm if max(1, it) < warm else 1.0)   # lr_scaler.py:18-28
    if torch_ddp:                                                                                 # :147-151, unchanged
        pipe.model = torch.nn.parallel.DistributedDataParallel(pipe.model, device_ids=[torch.cuda.current_device()])
    elif world > 1 or force_collectives:                                                          # ... or our wrapper in its place
        pipe.model = mdist.DataParallel(pipe.model, device_ids=[torch.cuda.current_device()], bucket_mb=bucket_mb,
                                        wire_dtype=wire, force_collectives=force_collectives, record_timeline=True)
    ema = trainer.ModelEma(vm)                                                                    # :157
    args = types.SimpleNamespace(fp16=bf16, gradient_clip_norm=2.0)
    scaler = torch.amp.GradScaler("cuda") if bf16 else None                                      # :113-116
    if plain:
        opt._mdm_fused = False
    if serial_wgrad:
        os.environ["MDM_HIP_SERIAL_WGRAD"] = "1"

    def step(sample):
        return trainer.train_batch(pipe, sample, opt, sched, None, args, grad_scaler=scaler, accumulate_gradient=False,
                                   num_grad_accumulations=1, ema_model=ema, loss_factor=1.0)[0]

    return step, opt


def _binary_provenance():
    """the library these numbers were measured on: path, SHA-256 of the loaded file, the git revision it was built from"""
    from mdm_hip import _lib
    return _lib.provenance()


def timed_steps(step, sample, warmup, steps, sync):
    for _ in range(warmup):
        step(sample)
    sync()
    # BENCH_STEP_TIMES=1 (diagnostic): a HIP event after every timed step, read back after the closing sync -- the
    # per-step GPU intervals go to stderr, the timed region itself is unchanged
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if os.environ.get("BENCH_STEP_TIMES") else None
    # The cyclic garbage collector stays out of the timed region (collected before, re-enabled after; reference counting
    # frees the step's tensors as always): a generation-2 pass over the module / autograd object graph is a 100+ ms host
    # stall at a random step.  (It was suspected of the occasional first run on a box reading 107-112 ms where the next
    # process reads 93-95; those still occur with the collector off.)
    gc.collect()
    gc.disable()
    try:
        t0 = time.perf_counter()
        if evs:
            evs[0].record()
        for i in range(steps):
            step(sample)
            if evs:
                evs[i + 1].record()
        sync()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()   # (an exception in a timed step -- an OOM caught by an outer leg -- must not leave the collector off)
    if evs:
        print("per-step ms: " + " ".join("%.1f" % evs[i].elapsed_time(evs[i + 1]) for i in range(steps)), file=sys.stderr)
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="unet64", choices=["unet64", "nested256", "mini"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 64 / 16)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--bucket-mb", type=float, default=256.0, help="gradient all-reduce bucket size (N > 1)")
    ap.add_argument("--wire-bf16", action="store_true", help="all-reduce gradients in bf16 (N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sampling", action="store_true")
    ap.add_argument("--serial-wgrad", action="store_true",
                    help="diagnostic: weight gradients on the main stream (uncontended per-kernel durations for profiling)")
    ap.add_argument("--no-nested", action="store_true", help="skip the nested256 (configs[2]) sub-measurement")
    ap.add_argument("--no-reference-loop", action="store_true", help="skip the plain-path (torch optimizer / EMA / autograd accumulation) leg")
    ap.add_argument("--reference-loop", action="store_true", help="run the plain-path leg also when N > 1 (torch DDP, as the unchanged CLI)")
    ap.add_argument("--no-nested1024", action="store_true", help="skip the nested-1024 (configs[4]) sampling leg")
    ap.add_argument("--wire-fp32", action="store_true", help="fp32 gradients on the wire (the default)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: run the gradient buckets through RCCL anyway (world-of-one process group)")
    ap.add_argument("--sample-batch", type=int, default=None)
    args = ap.parse_args()

    from mdm_hip import distributed as mdist
    from mdm_hip import ops

    if args.force_collectives and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
        dist.init_process_group(backend=os.environ.get("MDM_DIST_BACKEND", "nccl"), init_method="env://", world_size=1, rank=0)
    # MDM_BENCH_DEVICE=0 lets two ranks share ONE GPU (development check of the N > 1 path on the single-GPU box; timing
    # meaningless); the driver's real runs use RCCL with one GPU per rank.  With MDM_DIST_BACKEND=gloo the wire is gloo; with
    # RCCL each rank claims its own host id -- RCCL refuses two ranks of one host on one device, and then talks over the
    # socket transport on loopback (the recipe of tests/test_distributed_gpu.py::test_two_rank_train_step_over_rccl)
    if "MDM_BENCH_DEVICE" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("MDM_DIST_BACKEND", "nccl") == "nccl":
        os.environ.setdefault("NCCL_HOSTID", "mdm-bench-rank-%s" % os.environ.get("RANK", "0"))
        os.environ["LOCAL_RANK"] = os.environ["MDM_BENCH_DEVICE"]
        for k, v in (("NCCL_SOCKET_IFNAME", "lo"), ("NCCL_IB_DISABLE", "1"), ("NCCL_P2P_DISABLE", "1"), ("NCCL_SHM_DISABLE", "1")):
            os.environ.setdefault(k, v)
    local, rank, world = mdist.init_distributed_singlenode(backend=os.environ.get("MDM_DIST_BACKEND"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if "MDM_BENCH_DEVICE" in os.environ:
        local = int(os.environ["MDM_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    batch = args.batch or {"unet64": 64, "nested256": 16, "mini": 4}[args.workload]
    bf16 = args.dtype == "bf16"
    wire = torch.bfloat16 if args.wire_bf16 else (torch.float32 if args.wire_fp32 else "auto")   # auto: fp32 (bf16 on the wire is opt-in)

    def sync():
        if world > 1:
            mdist.barrier()
        torch.cuda.synchronize()

    pipe, side = build(args.workload, device)
    step, opt = make_step(pipe, bf16, world, bucket_mb=args.bucket_mb, wire=wire, serial_wgrad=args.serial_wgrad,
                          force_collectives=args.force_collectives)
    sample = synthetic_batch(batch, side, device, seed=1234 + rank)
    # Settle phase (disclosed in the line as config.settle_steps): untimed train steps ahead of the W warm-up steps.  This
    # leg is the first sustained GPU work of the process, after ~20 s of host-side model construction; four of this
    # round's ~25 default runs read 107-112 ms here while every later leg of the same process (nested-256, sampling) and
    # the next process on the same box read their usual figures -- a start-of-process transient of a second or two, not a
    # property of the step.  The timed region is untouched: W warm-up steps, barrier + synchronize, exactly K steps.
    settle = int(os.environ.get("BENCH_SETTLE_STEPS", "15"))
    for _ in range(settle):
        step(sample)
    dt = timed_steps(step, sample, args.warmup, args.steps, sync)
    assert getattr(opt, "_mdm_fused", False) not in (None, False), getattr(opt, "_mdm_fused_reason", "the fused path did not engage")
    # what went over the wire, and when: per bucket of the LAST step (bucket, MB, issued at, compute stream free at; ms from
    # the first bucket's issue) -- how much of the gradient exchange hid behind backward on this node
    comm_wire, comm_timeline, comm_window = "fp32", None, None
    red = getattr(pipe.model, "reducer", None)
    if red is not None:
        comm_wire = "bf16" if red.wire_dtype == torch.bfloat16 else "fp32"
        comm_timeline = [[b, round(nb / 1e6, 1), round(t0, 3), round(t1, 3)] for b, nb, t0, t1 in red.timeline()]
        win = red.backward_window_ms()
        if win is not None:   # where the bucket times (relative to the first issue) sit inside the backward pass
            comm_window = {"first_issue_ms_after_backward_start": round(win[0], 3), "backward_ms": round(win[1], 3)}
    if world > 1:
        tt = torch.tensor([dt], device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # secondary metric: sampling latency (replicas only: every rank samples its own prompts, no collective)
    samp = None
    if not args.no_sampling:
        n_it = 6
        pipe.eval()
        sb = args.sample_batch or batch
        ssample = synthetic_batch(sb, side, device, seed=4321 + rank)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            pipe.sample(sb, ssample, side, device, resample_steps=True, num_inference_steps=2, ddim_eta=0)  # warm-up
            sync()
            ts = time.perf_counter()
            pipe.sample(sb, ssample, side, device, resample_steps=True, num_inference_steps=n_it, ddim_eta=0)
            sync()
            dts = time.perf_counter() - ts
            ms_eager = dts / n_it * 1e3
            ms_it, how = ms_eager, "eager sampler"
            if world == 1:
                # the same iterations as ONE hipGraph replay each (mdm_hip.graph.GraphedSampler): the product's sampling
                # path.  Single-process runs only: a stream capture while RCCL's watchdog thread is polling events is
                # not something this bench should bet the headline line on.
                try:
                    from mdm_hip.graph import GraphedSampler
                    gs = GraphedSampler(pipe)
                    gs.sample(sb, ssample, side, device, num_inference_steps=n_it, ddim_eta=0)   # builds + warms the graph
                    sync()
                    ts = time.perf_counter()
                    gs.sample(sb, ssample, side, device, num_inference_steps=n_it, ddim_eta=0)
                    sync()
                    ms_it, how = (time.perf_counter() - ts) / n_it * 1e3, "one hipGraph replay per iteration (GraphedSampler)"
                    del gs
                except Exception as ex:   # secondary metric: never take the run down
                    how = "eager sampler (graph capture failed: %s)" % str(ex)[:80]
        demo_steps = 50 if args.workload == "unet64" else 100   # generate_sample.py:546-551 demo defaults
        samp = {"ms_per_denoise_step": round(ms_it, 3), "ms_per_denoise_step_eager": round(ms_eager, 3), "batch_per_gpu": sb,
                "images_per_s_at_%d_steps" % demo_steps: round(world * sb / (demo_steps * ms_it / 1e3), 3),
                "sampler": "DDIM eta=0, CFG off; " + how, "timed_steps": n_it,
                "precision_note": "bf16 autocast: 4-8e-3 rel-L2 from the fp32 reference over 25-100 steps (outside the 1e-3 gate); "
                                  "the reference samples in fp32 -- see the fp32 legs"}
        if bf16:
            # The reference's sampling path has no autocast (diffusion.py:181-197, clis/generate_sample.py:230-256): the
            # same iterations on fp32 tensors, (a) products as three bf16 MFMAs (MDM_F32_SPLIT -- passes the 1e-3 gate on
            # the long-horizon goldens, tests/test_model_gpu.py) and (b) the exact-fp32 MFMA path, for scale.
            def fp32_leg(split, iters, graphed=False):
                with torch.no_grad(), ops.fp32_split(split):
                    if graphed:   # the product's sampling path, in this arithmetic (the graph is keyed on it)
                        from mdm_hip.graph import GraphedSampler
                        g = GraphedSampler(pipe, warmup=1)
                        g.sample(sb, ssample, side, device, num_inference_steps=iters, ddim_eta=0)   # builds the graph
                        sync()
                        t0_ = time.perf_counter()
                        g.sample(sb, ssample, side, device, num_inference_steps=iters, ddim_eta=0)
                        sync()
                        return (time.perf_counter() - t0_) / iters * 1e3
                    pipe.sample(sb, ssample, side, device, resample_steps=True, num_inference_steps=1, ddim_eta=0)
                    sync()
                    t0_ = time.perf_counter()
                    pipe.sample(sb, ssample, side, device, resample_steps=True, num_inference_steps=iters, ddim_eta=0)
                    sync()
                    return (time.perf_counter() - t0_) / iters * 1e3
            try:
                ms_x3e = fp32_leg(True, 4)
                ms_x3, how3 = ms_x3e, "eager"
                if world == 1:
                    try:
                        ms_x3, how3 = fp32_leg(True, 4, graphed=True), "one hipGraph replay per iteration (GraphedSampler)"
                    except Exception as ex:
                        how3 = "eager (graph capture failed: %s)" % str(ex)[:80]
                samp["fp32_bf16x3"] = {"ms_per_denoise_step": round(ms_x3, 3), "ms_per_denoise_step_eager": round(ms_x3e, 3),
                                       "ratio_to_bf16": round(ms_x3 / ms_it, 2), "ratio_to_bf16_eager": round(ms_x3e / ms_eager, 2),
                                       "images_per_s_at_%d_steps" % demo_steps: round(world * sb / (demo_steps * ms_x3 / 1e3), 3),
                                       "arithmetic": "fp32 tensors / accumulators, products as 3 bf16 MFMAs on hi + lo halves; 1e-3 gate: pass",
                                       "sampler": how3, "timed_steps": 4}
                ms_ex = fp32_leg(False, 2)
                samp["fp32_exact"] = {"ms_per_denoise_step": round(ms_ex, 3), "ratio_to_bf16_eager": round(ms_ex / ms_eager, 2),
                                      "arithmetic": "v_mfma_f32_16x16x4_f32 (1/16 of the bf16 MFMA rate)", "sampler": "eager", "timed_steps": 2}
            except Exception as ex:
                samp["fp32_bf16x3"] = {"error": str(ex)[:200]}
    roof = None
    if not args.no_roofline:
        # one extra, untimed step with HIP events around every GEMM-class / streaming launch (on the launch stream).
        # Every rank runs the step (it contains the gradient all-reduce); only rank 0 records and reports.
        pipe.train()
        if rank == 0:
            ops.profile_begin()
        step(sample)
        torch.cuda.synchronize()
        if rank == 0:
            roof = ops.profile_end(PEAK_BF16_TFLOPS if bf16 else PEAK_F32_TFLOPS, PEAK_HBM_GBS)
            if roof is not None and args.workload == "unet64" and bf16 and batch == 64:
                roof["traffic"], roof["traffic_source"] = pmc_traffic(roof["kernel"])   # HBM bytes per launch (PMC), null if not collected
    if world > 1:
        mdist.barrier()

    # BASELINE.json configs[2]: the nested 64+256 train step at batch 16, same protocol, fewer steps
    nested = None
    if args.workload == "unet64" and not args.no_nested and bf16:
        del step, opt
        ops.set_grad_sink(None)
        pipe = sample = None
        torch.cuda.empty_cache()
        try:
            npipe, nside = build("nested256", device)
            nstep, nopt = make_step(npipe, True, world, bucket_mb=args.bucket_mb, wire=wire)
            nsample = synthetic_batch(16, nside, device, seed=99 + rank)
            nsteps = max(3, min(args.steps, 10))
            ndt = timed_steps(nstep, nsample, 4, nsteps, sync)   # 4 warm-up steps: the allocator re-grows after empty_cache()
            if world > 1:
                tt = torch.tensor([ndt], device=device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ndt = float(tt.item())
            nflop = 3 * FWD_GFLOP_PER_SAMPLE["nested256"] * 16 / 1e3
            assert getattr(nopt, "_mdm_fused", False) not in (None, False)
            nested = {"workload": "cc12m_256x256 NestedUNet (64+256) train step, bf16, per-GPU batch 16", "steps": nsteps,
                      "ms_per_step": round(ndt / nsteps * 1e3, 3), "steps_per_s_whole_job": round(world * nsteps / ndt, 4),
                      "step_algorithmic_tflop": round(nflop, 2),
                      "step_mfma_roofline_frac": round(nflop / (ndt / nsteps) / PEAK_BF16_TFLOPS, 4)}
        except Exception as ex:   # a secondary measurement never takes the headline line down (one process; with several
            if world > 1:         # ranks a local failure would leave the others in a collective, so it propagates)
                raise
            nested = {"error": str(ex)[:200]}

    # the same entry point on its PLAIN path: what the CLI's own objects do when nothing is adopted into arenas
    ref_loop = None
    if args.workload == "unet64" and bf16 and not args.no_reference_loop and (world == 1 or args.reference_loop):
        try:
            nstep = nopt = npipe = nsample = None
            ops.set_grad_sink(None)
            ops.enable_async_wgrad(False)
            ops.enable_deferred_wgrad(False)
            torch.cuda.empty_cache()
            rpipe, rside = build("unet64", device)
            rstep, ropt = make_step(rpipe, True, world, plain=True, torch_ddp=world > 1)
            rsample = synthetic_batch(batch, rside, device, seed=1234 + rank)
            rsteps = max(2, min(args.steps, 4))
            rdt = timed_steps(rstep, rsample, 1, rsteps, sync)
            assert getattr(ropt, "_mdm_fused", None) is False
            ref_loop = {"ms_per_step": round(rdt / rsteps * 1e3, 3), "steps_per_s_whole_job": round(world * rsteps / rdt, 4), "steps": rsteps,
                        "what": "mdm_hip.trainer.train_batch, plain path: HIP denoiser kernels, but autograd gradient accumulation, "
                                "GradScaler, clip_grad_norm_, torch.optim.AdamW, per-tensor ModelEma.update"
                                + (", torch DistributedDataParallel" if world > 1 else "")}
            del rstep, ropt, rpipe
            torch.cuda.empty_cache()
        except Exception as ex:   # secondary leg: never take the headline line down
            ref_loop = {"error": str(ex)[:200]}

    # BASELINE.json configs[4]: nested 64+256+1024 sampling, ancestral DDPM, batch 4 (generate_sample.py:546-551), bf16
    n1024 = None
    if args.workload == "unet64" and bf16 and world == 1 and not args.no_nested1024 and not args.no_sampling:
        try:
            import mdm_hip
            from mdm_hip import configs, diffusion, samplers
            from mdm_hip.graph import GraphedSampler
            from mdm_hip.testing import randomize_zero_params

            sc = samplers.SamplerConfig(num_diffusion_steps=1000, schedule_type="DEEPFLOYD", prediction_type="V_PREDICTION",
                                        loss_target_type="DDPM", schedule_shifted=True, rescale_signal=1, schedule_shifted_power=2)
            torch.manual_seed(0)
            net = mdm_hip.NestedUNet(3, 3, configs.nested1024_config(2048))
            net.load_state_dict(randomize_zero_params(net.state_dict(), seed=1))
            p4 = diffusion.NestedDiffusion(net, diffusion.NestedDiffusionConfig(sampler_config=sc, use_vdm_loss_weights=False,
                                                                                use_double_loss=True, no_use_residual=True)).to(device)
            s4 = synthetic_batch(4, 64, device, seed=7)
            n_it = 6
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                gs = GraphedSampler(p4, seed=7)
                gs.sample(4, s4, 1024, device, num_inference_steps=n_it, ddim_eta=1)
                sync()
                ts = time.perf_counter()
                out4 = gs.sample(4, s4, 1024, device, num_inference_steps=n_it, ddim_eta=1)
                sync()
                ms4 = (time.perf_counter() - ts) / n_it * 1e3
            n1024 = {"workload": "flickr1024 NestedUNet (64+256+1024) DDPM sampling (ddim_eta=1), batch 4, bf16, synthetic checkpoint",
                     "ms_per_denoise_step": round(ms4, 3), "timed_steps": n_it, "seconds_per_250_steps": round(ms4 * 250 / 1e3, 3),
                     "images_per_s_at_250_steps": round(4 / (ms4 * 250 / 1e3), 4), "finite": bool(torch.isfinite(out4).all()),
                     "alg_tflop_per_step": round(4 * 1018.6 / 1e3, 3),
                     "mfma_roofline_frac": round(4 * 1018.6e9 / (ms4 / 1e3) / (PEAK_BF16_TFLOPS * 1e12), 4),
                     "sampler": "one hipGraph replay per iteration (GraphedSampler), CFG off"}
            # the 32 / 64-channel outer levels of this model are under the MFMA ridge -- judge the leg
            # against HBM as well.  Bytes of one denoise iteration from the PMC counters (separate FETCH_SIZE / WRITE_SIZE
            # passes, tools/sample_pmc.py); the time is this run's.
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_nested1024_sampling.json")))
                hb = float(pm["hbm_bytes_per_iteration"])
                n1024["hbm"] = {"bytes_per_denoise_step": int(hb), "gb_per_s": round(hb / (ms4 / 1e3) / 1e9, 1),
                                "frac_of_peak": round(hb / (ms4 / 1e3) / (PEAK_HBM_GBS * 1e9), 4), "peak_gb_per_s": PEAK_HBM_GBS,
                                "traffic_source": "profiles/r06_pmc_nested1024_sampling.json (PMC, eager sampler; the graphed replay launches the same kernels)"}
            except (OSError, KeyError, ValueError):
                n1024["hbm"] = None
            del gs
            # ... and at the reference's precision class: fp32 tensors, bf16x3 products (the configs[4] golden passes the
            # 1e-3 gate in this mode, tests/test_model_gpu.py::test_long_horizon_sampling_matches_reference)
            try:
                with torch.no_grad(), ops.fp32_split(True):
                    p4.sample(4, s4, 1024, device, resample_steps=True, num_inference_steps=1, ddim_eta=1)
                    sync()
                    ts = time.perf_counter()
                    p4.sample(4, s4, 1024, device, resample_steps=True, num_inference_steps=3, ddim_eta=1)
                    sync()
                    ms4e = (time.perf_counter() - ts) / 3 * 1e3
                    ms4x, how4 = ms4e, "eager"
                    try:
                        g4 = GraphedSampler(p4, warmup=1)
                        g4.sample(4, s4, 1024, device, num_inference_steps=3, ddim_eta=1)   # builds the graph
                        sync()
                        ts = time.perf_counter()
                        g4.sample(4, s4, 1024, device, num_inference_steps=3, ddim_eta=1)
                        sync()
                        ms4x, how4 = (time.perf_counter() - ts) / 3 * 1e3, "one hipGraph replay per iteration (GraphedSampler)"
                        del g4
                    except Exception as ex:
                        how4 = "eager (graph capture failed: %s)" % str(ex)[:80]
                n1024["fp32_bf16x3"] = {"ms_per_denoise_step": round(ms4x, 3), "ms_per_denoise_step_eager": round(ms4e, 3),
                                        "ratio_to_bf16": round(ms4x / ms4, 2), "seconds_per_250_steps": round(ms4x * 250 / 1e3, 3),
                                        "sampler": how4, "timed_steps": 3}
            except Exception as ex:
                n1024["fp32_bf16x3"] = {"error": str(ex)[:200]}
            del p4, net
            torch.cuda.empty_cache()
        except Exception as ex:
            n1024 = {"error": str(ex)[:200]}

    if rank == 0:
        ms = dt / args.steps * 1e3
        steps_per_s = world * args.steps / dt
        alg_tflop_step = 3 * FWD_GFLOP_PER_SAMPLE[args.workload] * batch / 1e3
        out = {
            "metric": "denoise-steps/sec, whole job = N x optimizer steps/s at per-GPU batch %d (train fwd+bwd+optimizer, %s)" % (
                batch, "cc12m_64x64 U-Net" if args.workload == "unet64" else "cc12m_256x256 NestedUNet"),
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
                "settle_steps": settle,
                "optimizer_steps_per_s": round(args.steps / dt, 4),
                "samples_per_s": round(steps_per_s * batch, 2),
                "step_algorithmic_tflop": round(alg_tflop_step, 2),
                "step_mfma_roofline_frac": round(alg_tflop_step / (ms / 1e3) / (PEAK_BF16_TFLOPS if bf16 else PEAK_F32_TFLOPS), 4),
                "comm": {"backend": dist.get_backend() if dist.is_initialized() else None, "world_size": world, "forced_collectives": bool(args.force_collectives),
                         "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.is_initialized() else None,
                         "bucket_mb": args.bucket_mb, "wire_dtype": comm_wire, "bucket_timeline_ms": comm_timeline, "backward_window": comm_window},
            },
            "roofline": roof,
            "sampling": samp,
            "nested256": nested,
            "reference_loop": ref_loop,
            "nested1024_sampling": n1024,
            "entry_point": "mdm_hip.trainer.train_batch (reference trainer.py:13-25 signature) on torch AdamW + LambdaLR + ModelEma objects, fused arena path",
            "binary": _binary_provenance(),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        mdist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

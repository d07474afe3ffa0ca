"""GPU: ``mdm_hip.trainer.train_batch`` (the reference's entry point, trainer.py:13-96) on the real denoiser.
  * fused path (flat arenas, gradient sink, mdm_sumsq + mdm_adamw_ema_step) == plain path (autograd accumulation,
    clip_grad_norm_, torch AdamW, ModelEma.update) over optimizer steps with gradient accumulation, a warm-up
    scheduler and EMA warm-up -- parameters, EMA, Adam moments, returned losses, logger rows;
  * what the CLI does with the objects afterwards keeps working: optimizer.state_dict(), ema_model.save / load,
    vision_model.save / load (train_parallel.py:270-293);
  * two ranks sharing the GPU (gloo): torch's DistributedDataParallel around ``diffusion_model.model`` exactly as
    train_parallel.py:147-154 wraps it (plain path), and mdm_hip.distributed.DataParallel in its place (fused path),
    each against one process on the concatenated batch."""
import os
import socket
import sys
import types
from contextlib import nullcontext

import pytest
import torch
import torch.multiprocessing as mp

import make_golden as MG
import parity_cases as PC
import unet_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pipe(name="mini_unet"):
    from mdm_hip import diffusion as D
    from mdm_hip import samplers as S

    model, _, _ = PC.build_module(name)
    scfg = S.SamplerConfig(num_diffusion_steps=1000, schedule_type="DEEPFLOYD", prediction_type="V_PREDICTION", loss_target_type="DDPM")
    return D.Diffusion(model, D.DiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False)).to(torch.device("cuda:0"))


def _sample(n=2):
    inp = PC.inputs("mini_unet")
    g = torch.Generator().manual_seed(29)
    return {"lm_outputs": inp["cond"][:n].cuda(), "lm_mask": inp["mask"][:n].cuda(),
            "images": (torch.rand(2, 3, 16, 16, generator=g) * 2 - 1)[:n].cuda()}


def _drive(mode, fp16=False, n_micro=6, accumulations=2, wrap=None, sample=None, loss_kw=None, lr=1e-3):
    """the loop of clis/train_parallel.py:122-230 around train_batch; mode: "fused" | "plain" """
    from mdm_hip import ops, trainer

    ops.set_grad_sink(None)
    pipe = _pipe()
    vm = pipe.model.vision_model
    opt = torch.optim.AdamW(vm.parameters(), lr=lr, weight_decay=0, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: min(1.0, (it + 1) / 3))
    if wrap is not None:
        pipe.model = wrap(pipe.model)
    ema = trainer.ModelEma(vm, decay=0.9, warmup_steps=1)
    if mode == "plain":
        opt._mdm_fused = False
    logger = MG.RecordingLogger()
    args = types.SimpleNamespace(fp16=fp16, gradient_clip_norm=0.5)
    sample = sample or _sample()
    if loss_kw:   # fixed timesteps / noise (two-rank comparisons)
        orig = pipe.get_loss
        pipe.get_loss = lambda s: orig(s, **loss_kw)
    vals = []
    for i, acc in enumerate(MG.trainer_schedule(n_micro, accumulations)):
        torch.manual_seed(100 + i)
        ctx = pipe.model.no_sync() if (acc and hasattr(pipe.model, "no_sync")) else nullcontext()
        with ctx:
            out = trainer.train_batch(pipe, sample, opt, sched, logger, args, grad_scaler=None, accumulate_gradient=acc,
                                      num_grad_accumulations=accumulations, ema_model=ema, loss_factor=1.0)
        vals.append(out[0])
    torch.cuda.synchronize()
    fused = getattr(opt, "_mdm_fused", None)
    assert (fused not in (None, False)) == (mode == "fused"), getattr(opt, "_mdm_fused_reason", None)
    cpu = lambda t: t.detach().float().cpu().clone()
    res = {"loss": vals, "log": logger.rows,
           "p": {k: cpu(v) for k, v in vm.named_parameters()},
           "ema": {k: cpu(v) for k, v in ema.module.named_parameters()},
           "m": {k: cpu(opt.state[v]["exp_avg"]) for k, v in vm.named_parameters()}}
    ops.set_grad_sink(None)
    return res, (pipe, opt, ema)


def _agg(a, b):
    num = sum(float((a[k].double() - b[k].double()).pow(2).sum()) for k in b)
    den = sum(float(b[k].double().pow(2).sum()) for k in b)
    return (num / den) ** 0.5


def test_fused_train_batch_matches_plain_path():
    plain, _ = _drive("plain")
    fused, _ = _drive("fused")
    assert all(abs(a - b) <= 1e-4 * abs(b) + 1e-7 for a, b in zip(fused["loss"], plain["loss"])), (fused["loss"], plain["loss"])
    assert [r[0] for r in fused["log"]] == [r[0] for r in plain["log"]] and len(fused["log"]) == 6
    assert all(abs(a[1] - b[1]) <= 1e-4 * abs(b[1]) + 1e-7 for a, b in zip(fused["log"], plain["log"]))
    # Adam normalises every gradient to +-lr, noise gradients included: aggregate comparison (see test_model_gpu.py)
    assert _agg(fused["m"], plain["m"]) < 1e-4
    assert _agg(fused["p"], plain["p"]) < 1e-4 and _agg(fused["ema"], plain["ema"]) < 1e-4
    assert _agg(fused["ema"], fused["p"]) > 1e-6    # the EMA lags: it is its own arena, not an alias


def test_fused_train_batch_bf16_runs_and_tracks_plain_path():
    """args.fp16 = True (bf16 autocast, trainer.py:29-30): one optimizer step; Adam's first moment after one step is
    (1 - beta1) x the clipped gradient -- linear in what backward produced"""
    plain, _ = _drive("plain", fp16=True, n_micro=2)
    fused, _ = _drive("fused", fp16=True, n_micro=2)
    assert _agg(fused["m"], plain["m"]) < 2e-2


def test_objects_the_cli_keeps_using_still_work(tmp_path):
    """after adoption the optimizer / EMA / model objects hold views of the arenas: their state_dict()s and checkpoint
    files must keep working (train_parallel.py:270-293)"""
    from mdm_hip import trainer

    res, (pipe, opt, ema) = _drive("fused", n_micro=4)
    vm = pipe.model.vision_model
    sd = opt.state_dict()
    assert len(sd["state"]) == len(list(vm.parameters()))
    k0 = next(iter(sd["state"]))
    assert float(sd["state"][k0]["exp_avg"].abs().sum()) > 0
    f_ema, f_vm = str(tmp_path / "vis_model_000002.pth"), str(tmp_path / "vis_model_noema_000002.pth")
    ema.save(f_ema, other_items={"batch_num": 2})
    vm.save(f_vm, other_items={"batch_num": 2})
    fresh, _, _ = PC.build_module("mini_unet", seed=5)
    assert fresh.load(f_vm)["batch_num"] == 2
    for (k, a), (_, b) in zip(fresh.state_dict().items(), vm.state_dict().items()):
        assert torch.equal(a, b.cpu()), k
    ema2 = trainer.ModelEma(fresh)
    ema2.load(f_ema)
    for (k, a), (_, b) in zip(ema2.module.state_dict().items(), ema.module.state_dict().items()):
        assert torch.equal(a, b.cpu()), k
    # a resumed optimizer state is taken over: second adoption starts from the loaded moments
    pipe2 = _pipe()
    vm2 = pipe2.model.vision_model
    vm2.load(f_vm)
    opt2 = torch.optim.AdamW(vm2.parameters(), lr=1e-3, weight_decay=0, eps=1e-8)
    opt2.load_state_dict(sd)
    st = trainer._adopt(pipe2, opt2, None)
    assert st is not None and int(st.step_dev) == 2
    p0 = next(iter(vm2.parameters()))
    assert torch.equal(opt2.state[p0]["exp_avg"].cpu(), opt.state[next(iter(vm.parameters()))]["exp_avg"].cpu())
    from mdm_hip import ops
    ops.set_grad_sink(None)


def test_adoption_guards_ema_and_reloaded_optimizer_state():
    """(advisor, round 3) the adoption is cached on the optimizer: a ModelEma that appears (or changes) later must not be
    silently ignored, and optimizer.load_state_dict() AFTER adoption must reach the arenas the fused kernel reads"""
    from mdm_hip import ops, trainer

    res, (pipe, opt, ema) = _drive("fused", n_micro=4)
    vm = pipe.model.vision_model
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 1.0)
    args = types.SimpleNamespace(fp16=False, gradient_clip_norm=0.5)
    with pytest.raises(RuntimeError, match="same ema_model"):
        trainer.train_batch(pipe, _sample(), opt, sched, None, args, ema_model=None)
    with pytest.raises(RuntimeError, match="same ema_model"):
        trainer.train_batch(pipe, _sample(), opt, sched, None, args, ema_model=trainer.ModelEma(vm))
    # load a state dict with recognisable moments: the views are re-installed and the arenas hold the loaded values
    st = opt._mdm_fused
    sd = opt.state_dict()
    for ent in sd["state"].values():
        ent["exp_avg"] = torch.full_like(ent["exp_avg"], 0.25)
        ent["exp_avg_sq"] = torch.full_like(ent["exp_avg_sq"], 0.5)
        ent["step"] = torch.tensor(7.0)
    opt.load_state_dict(sd)
    assert float(st.m.min()) == 0.25 and float(st.m.max()) == 0.25 and float(st.v.min()) == 0.5
    assert int(st.step_dev) == 7
    p0 = next(iter(vm.parameters()))
    assert opt.state[p0]["exp_avg"].data_ptr() == st.view(st.m, p0).data_ptr()
    assert float(opt.state_dict()["state"][0]["step"]) == 7.0
    trainer.train_batch(pipe, _sample(), opt, sched, None, args, ema_model=ema)   # and the step still runs on them
    torch.cuda.synchronize()
    assert int(st.step_dev) == 8 and float(st.m.max()) < 0.25 + 1e-6 and float((st.m - 0.25).abs().max()) > 0
    ops.set_grad_sink(None)


# ---- two ranks on one GPU ------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _four():
    inp = PC.inputs("mini_unet")
    g = torch.Generator().manual_seed(77)
    cond = torch.randn(4, 8, 64, generator=g)
    return {"images": torch.rand(4, 3, 16, 16, generator=g) * 2 - 1, "lm_outputs": cond, "lm_mask": torch.ones(4, 8),
            "time": torch.tensor([10, 300, 650, 990]), "noise": torch.randn(4, 3, 16, 16, generator=g)}


def _rank_run(mode, sel, wrap):
    b = _four()
    smp = {k: b[k][sel].cuda() for k in ("images", "lm_outputs", "lm_mask")}
    noise = b["noise"][sel].cuda()
    res, _ = _drive(mode, n_micro=4, accumulations=2, wrap=wrap, sample=smp,
                    loss_kw=dict(time=b["time"][sel].cuda(), noise_fn=lambda like: noise))
    return res


def _worker(rank, world, port, out, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    for p in (os.path.join(ROOT, "ml-mdm_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from mdm_hip import distributed as md

    md.init_distributed_singlenode(backend="gloo")
    if kind == "torch_ddp":   # reference clis/train_parallel.py:147-151
        wrap = lambda m: torch.nn.parallel.DistributedDataParallel(m, device_ids=[0])
        mode = "plain"
    else:
        wrap = lambda m: md.DataParallel(m, device_ids=[0], bucket_mb=0.25)
        mode = "fused"
    res = _rank_run(mode, slice(rank * 2, rank * 2 + 2), wrap)
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["torch_ddp", "mdm_data_parallel"])
def test_two_ranks_wrapped_like_the_cli_match_one_process(tmp_path, kind):
    """2 ranks x 2 samples, 2 optimizer steps of 2 accumulation micro-steps each (the first under no_sync), the model
    wrapped as train_parallel.py:147-154 wraps it == one process x 4 samples"""
    out = str(tmp_path / "r.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, kind), nprocs=2, join=True)
    two = torch.load(out, weights_only=False)
    one = _rank_run("fused", slice(0, 4), None)
    assert _agg(two["m"], one["m"]) < 1e-4
    assert _agg(two["p"], one["p"]) < 1e-4 and _agg(two["ema"], one["ema"]) < 1e-4


def _nan_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    for p in (os.path.join(ROOT, "ml-mdm_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from mdm_hip import distributed as md
    from mdm_hip import ops, trainer

    md.init_distributed_singlenode(backend="gloo")
    ops.set_grad_sink(None)
    pipe = _pipe()
    vm = pipe.model.vision_model
    opt = torch.optim.AdamW(vm.parameters(), lr=1e-3, weight_decay=0, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: min(1.0, (it + 1) / 4))
    pipe.model = md.DataParallel(pipe.model, device_ids=[0], bucket_mb=0.25)
    ema = trainer.ModelEma(vm, decay=0.9, warmup_steps=2)
    args = types.SimpleNamespace(fp16=True, gradient_clip_norm=0.5)
    b = _four()
    smp = {k: b[k][rank * 2:rank * 2 + 2].cuda() for k in ("images", "lm_outputs", "lm_mask")}
    vals = []
    for i in range(3):
        s_i = dict(smp)
        if i == 1 and rank == 1:   # only rank 1 sees a NaN loss, in the second step
            s_i["images"] = smp["images"].clone()
            s_i["images"][0, 0, 0, 0] = float("nan")
        torch.manual_seed(100 + i)
        vals.append(trainer.train_batch(pipe, s_i, opt, sched, None, args, ema_model=ema)[0])
    torch.cuda.synchronize()
    st = opt._mdm_fused
    # a step's skip flag is examined two calls later (round 6: one call later made the host wait for the whole previous
    # step) -- or when the optimizer's state is taken, as a checkpoint does: that is the settled view compared below
    opt.state_dict()
    res = {"loss": vals, "lr": sched.get_last_lr()[0], "ema_counter": ema.counter, "step": int(st.step_dev),
           "p": st.flat_p.detach().cpu().clone()}
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        torch.save(gathered, out)
    dist.barrier()
    dist.destroy_process_group()


def test_nan_on_one_rank_keeps_the_ranks_in_lockstep(tmp_path):
    """(advisor, rounds 3 and 4) bf16 fused step on two ranks, a NaN loss on rank 1 only: both ranks must skip the update
    (device side), keep identical parameters, and treat scheduler and EMA counter identically -- the reference's per-rank
    early return (trainer.py:38-41) would deadlock DDP.  Both ranks learn of the skip from the norm of the REDUCED gradient
    -- two calls later, or when optimizer.state_dict() is taken (no end-of-step wait: the host keeps its lead over the
    GPU) -- and then do what the
    reference's NaN branch does (bf16: no scheduler.step(), no EMA-counter increment), so the learning rate schedule and
    the EMA warm-up do not depend on the world size."""
    import math

    out = str(tmp_path / "r.pt")
    mp.spawn(_nan_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out, weights_only=False)
    assert math.isnan(r1["loss"][1]) and not math.isnan(r0["loss"][1])
    assert r0["lr"] == r1["lr"] and abs(r0["lr"] - 0.75e-3) < 1e-9           # two of the three steps counted: (2 + 1) / 4
    assert r0["ema_counter"] == r1["ema_counter"] == 2
    assert r0["step"] == r1["step"] == 2          # the NaN step did not advance the optimizer
    assert torch.equal(r0["p"], r1["p"]) and bool(torch.isfinite(r0["p"]).all())

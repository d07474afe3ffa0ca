"""GPU: the N > 1 train-step path end to end.  Two ranks share the single GPU of the test box (gloo moves the
buckets through the host; the driver's real multi-GPU runs use RCCL): gradient sink + side stream + bucket hooks +
fused optimizer must give the same parameters as one process training on the concatenated batch."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_paths():
    for p in (os.path.join(ROOT, "ml-mdm_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _batch():
    g = torch.Generator().manual_seed(77)
    return {
        "images": torch.rand(4, 3, 16, 16, generator=g) * 2 - 1,
        "lm_outputs": torch.randn(4, 8, 64, generator=g),
        "lm_mask": torch.ones(4, 8),
        "time": torch.tensor([10, 300, 650, 990]),
        "noise": torch.randn(4, 3, 16, 16, generator=g),
    }


def _run_step(sel, steps=2):
    _setup_paths()
    import parity_cases as PC
    from mdm_hip import diffusion as D
    from mdm_hip import ops
    from mdm_hip import samplers as S
    from mdm_hip.trainer import TrainStep

    ops.set_grad_sink(None)
    model, _, _ = PC.build_module("mini_unet")
    scfg = S.SamplerConfig(num_diffusion_steps=1000, schedule_type="DEEPFLOYD", prediction_type="V_PREDICTION",
                           loss_target_type="DDPM")
    pipe = D.Diffusion(model, D.DiffusionConfig(sampler_config=scfg, use_vdm_loss_weights=False)).to(torch.device("cuda:0"))
    step = TrainStep(pipe, bf16=False, lr=1e-3, clip_norm=1e9, fused=True, bucket_mb=4.0)
    b = _batch()
    smp = {k: b[k][sel].cuda() for k in ("images", "lm_outputs", "lm_mask")}
    noise = b["noise"][sel].cuda()
    for _ in range(steps):
        step(smp, time=b["time"][sel].cuda(), noise_fn=lambda like: noise)
    torch.cuda.synchronize()
    ops.set_grad_sink(None)
    return {k: v.detach().float().cpu().clone() for k, v in model.named_parameters()}


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    _setup_paths()
    import torch.distributed as dist
    from mdm_hip import distributed as md

    md.init_distributed_singlenode(backend="gloo")
    params = _run_step(slice(rank * 2, rank * 2 + 2))
    if rank == 0:
        torch.save(params, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_step_matches_single_process(tmp_path):
    out = str(tmp_path / "p.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    two = torch.load(out)
    one = _run_step(slice(0, 4))
    num = sum(float((two[k].double() - one[k].double()).pow(2).sum()) for k in one)
    den = sum(float(one[k].double().pow(2).sum()) for k in one)
    assert (num / den) ** 0.5 < 1e-4

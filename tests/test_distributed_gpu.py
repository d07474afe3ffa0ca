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


def _batch(side=16):
    g = torch.Generator().manual_seed(77)
    return {
        "images": torch.rand(4, 3, side, side, generator=g) * 2 - 1,
        "lm_outputs": torch.randn(4, 8, 64, generator=g),
        "lm_mask": torch.ones(4, 8),
        "time": torch.tensor([10, 300, 650, 990]),
        "noise": torch.randn(4, 3, side, side, generator=g),
    }


def _run_step(sel, steps=2, side=16, bf16=False, async_wgrad=True):
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
    wire = torch.bfloat16 if os.environ.get("MDM_HIP_TEST_WIRE") == "bf16" else "auto"
    step = TrainStep(pipe, bf16=bf16, lr=1e-3, clip_norm=1e9, fused=True, bucket_mb=0.25, async_wgrad=async_wgrad, wire_dtype=wire)
    b = _batch(side)
    smp = {k: b[k][sel].cuda() for k in ("images", "lm_outputs", "lm_mask")}
    noise = b["noise"][sel].cuda()
    for _ in range(steps):
        step(smp, time=b["time"][sel].cuda(), noise_fn=lambda like: noise)
    torch.cuda.synchronize()
    ops.set_grad_sink(None)
    if bf16:
        # Adam's first step is lr * sign(g): parameters after it magnify bf16 rounding noise wherever g ~ 0.  The first
        # moment after ONE step is (1 - beta1) * (rank-averaged gradient): linear in what the reducer produced.
        return {"m": step.m.detach().float().cpu().clone()}
    return {k: v.detach().float().cpu().clone() for k, v in model.named_parameters()}


def _worker(rank, world, port, out, side, bf16, async_wgrad=True, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    if backend == "nccl":
        # RCCL with a REAL peer on the one GPU of the test box.  Two ranks on one device fail RCCL's "Duplicate GPU detected"
        # check, which compares bus ids among ranks of the same HOST (there is no switch for it in this build: round 6 looked
        # through librccl.so's parameters) -- so each rank claims its own host id and the two talk over the socket transport
        # on the loopback interface.  Slow, but every piece of the reducer's GPU path runs against a peer: bucket hand-off
        # from the hooks, the communication / completion streams, work.wait() on a foreign stream, ncclAvg, the bf16 wire.
        os.environ.update(NCCL_HOSTID="mdm-test-host-%d" % rank, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1",
                          NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    _setup_paths()
    import torch.distributed as dist
    from mdm_hip import distributed as md

    try:
        md.init_distributed_singlenode(backend=backend, timeout=120)
        if backend == "nccl":
            t = torch.ones(4, device="cuda:0") * (rank + 1)
            dist.all_reduce(t)
            assert float(t[0]) == 3.0
    except Exception as e:   # noqa: BLE001  (an RCCL that cannot form this communicator: reported, not a failure of ours)
        if backend != "nccl":
            raise
        if rank == 0:
            torch.save({"__skip__": "%s: %s" % (type(e).__name__, str(e)[:300])}, out)
        return
    params = _run_step(slice(rank * 2, rank * 2 + 2), steps=1 if bf16 else 2, side=side, bf16=bf16, async_wgrad=async_wgrad)
    if rank == 0:
        torch.save(params, out)
    dist.barrier()
    dist.destroy_process_group()


def _rel(a, b):
    num = sum(float((a[k].double() - b[k].double()).pow(2).sum()) for k in b)
    den = sum(float(b[k].double().pow(2).sum()) for k in b)
    return (num / den) ** 0.5


def test_two_rank_train_step_matches_single_process(tmp_path):
    """fp32: two ranks x 2 samples == one process x 4 samples (parameters after two optimizer steps).  The GroupNorm
    parameter gradients take the deferred route (per-sample rows + one reduce per flush point, reported to the
    reducer after their autograd node returned), in many small buckets."""
    out = str(tmp_path / "p.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, 16, False), nprocs=2, join=True)
    two = torch.load(out)
    one = _run_step(slice(0, 4))
    assert _rel(two, one) < 1e-4


def test_two_rank_deferred_weight_gradients_bf16(tmp_path):
    """bf16 at 128 x 128: 2 x 64 x 64 = 8192 pixels per rank at the attention level, so the 1x1 weight gradients take
    the QUEUED route (launched at flush points, reported after their autograd node returned).  A bucket released by
    autograd's own hook before those gradients were written would change the reduced gradient; compared with the same
    two-rank step without side stream / queue / deferral (same batch split, so bf16 rounding is the same -- against a
    single process the batch-size-dependent reduction orders alone move a bf16 gradient by ~1e-2)."""
    res = []
    for async_wgrad in (True, False):
        out = str(tmp_path / ("p%d.pt" % async_wgrad))
        mp.spawn(_worker, args=(2, _free_port(), out, 128, True, async_wgrad), nprocs=2, join=True)
        res.append(torch.load(out))
    assert _rel(res[0], res[1]) < 1e-4


@pytest.mark.parametrize("bf16_wire", [False, True])
def test_two_rank_train_step_over_rccl(tmp_path, bf16_wire):
    """The same two-rank step with backend="nccl" (RCCL) and a real peer -- see _worker for how two ranks share the test
    box's single GPU.  fp32 wire: parameters after two optimizer steps == one process on the concatenated batch (1e-4);
    bf16 wire (MDM_HIP_WIRE=bf16: division before the cast, copy-back on the completion stream): within bf16 rounding of it.
    Skipped with RCCL's own message when this build cannot form the communicator."""
    out = str(tmp_path / "p.pt")
    if bf16_wire:
        os.environ["MDM_HIP_TEST_WIRE"] = "bf16"
    try:
        mp.spawn(_worker, args=(2, _free_port(), out, 16, False, True, "nccl"), nprocs=2, join=True)
    finally:
        os.environ.pop("MDM_HIP_TEST_WIRE", None)
    two = torch.load(out)
    if "__skip__" in two:
        pytest.skip("RCCL could not form a two-rank communicator on one GPU: " + two["__skip__"])
    one = _run_step(slice(0, 4))
    assert _rel(two, one) < (2e-2 if bf16_wire else 1e-4)

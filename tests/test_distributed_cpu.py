"""CPU, world_size 2 over gloo: the gradient reducer (mdm_hip.distributed.GradReducer) averages
gradients across ranks, overlaps the bucket all-reduces with backward via post-accumulate hooks,
and matches single-process full-batch training; no_sync() suppresses communication."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed=0):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(16, 64), nn.SiLU(), nn.Linear(64, 64), nn.SiLU(), nn.Linear(64, 8))


def _worker(rank, world, port, bucket_mb, wire, out):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-mdm_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mdm_hip import distributed as md

    local, grank, w = md.init_distributed_singlenode(backend="gloo")
    assert (grank, w) == (rank, world)
    model = _model(seed=rank)  # different init per rank: broadcast must fix it
    red = md.GradReducer(list(model.parameters()), bucket_mb=bucket_mb, wire_dtype=wire)
    red.broadcast_parameters(0)
    g = torch.Generator().manual_seed(100)
    x_all, y_all = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    xs, ys = x_all[rank::world], y_all[rank::world]
    # micro-step under no_sync (accumulate only), then a synchronised step
    with red.no_sync():
        ((model(xs) - ys) ** 2).mean().backward()
    red.finish()
    local_only = red.flat.clone()
    ((model(xs) - ys) ** 2).mean().backward()
    red.finish()
    if rank == 0:
        torch.save({"flat": red.flat.clone(), "local_only": local_only, "nb": len(red.buckets),
                    "w0": model[0].weight.detach().clone()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb,wire", [(256.0, None), (0.004, None), (0.004, torch.bfloat16)])
def test_grad_reducer_gloo_world2(tmp_path, bucket_mb, wire):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), bucket_mb, wire, out), nprocs=2, join=True)
    got = torch.load(out)
    # reference: single process, rank-0 weights, per-rank losses averaged
    model = _model(seed=0)
    assert torch.equal(got["w0"], model[0].weight.detach())
    g = torch.Generator().manual_seed(100)
    x_all, y_all = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    params = list(model.parameters())

    def flat_grad(loss):
        gs = torch.autograd.grad(loss, params)
        return torch.cat([t.reshape(-1) for t in reversed(gs)])

    l0 = ((model(x_all[0::2]) - y_all[0::2]) ** 2).mean()
    l1 = ((model(x_all[1::2]) - y_all[1::2]) ** 2).mean()
    g0, g1 = flat_grad(l0), flat_grad(l1)
    assert torch.allclose(got["local_only"], g0, atol=1e-6)              # no_sync: purely local
    # second backward accumulated on top of the local micro-step, then everything was averaged
    expect = (2 * g0 + 2 * g1) / 2
    tol = 1e-5 if wire is None else 2e-2
    assert torch.allclose(got["flat"], expect, atol=tol * float(expect.abs().max())), float((got["flat"] - expect).abs().max())
    assert got["nb"] == (1 if bucket_mb > 1 else got["nb"]) and got["nb"] >= 1
    if bucket_mb < 1:
        assert got["nb"] > 1  # several buckets were in flight during backward


class _SinkLinear(torch.autograd.Function):
    """stands in for the HIP ops under a gradient sink: the weight / bias gradients are ADDED into the reducer's arena
    slots from inside backward, reported with ready(), and autograd gets None back (mdm_hip.ops.ConvFn.backward)"""
    sink = None
    deferred = False     # queue the gradient and report it after backward (grouped weight gradients, GroupNorm sums)
    queue = []

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w, b)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        s = _SinkLinear.sink
        if _SinkLinear.deferred:
            s.defer(w)
            s.defer(b)
            _SinkLinear.queue.append((w, b, dy.t() @ x, dy.sum(0)))
            return dy @ w, None, None
        s.slot(w).add_(dy.t() @ x)
        s.slot(b).add_(dy.sum(0))
        s.ready(w)
        s.ready(b)
        return dy @ w, None, None

    @staticmethod
    def flush():
        s = _SinkLinear.sink
        for w, b, dw, db in _SinkLinear.queue:
            s.slot(w).add_(dw)
            s.slot(b).add_(db)
            s.ready(w)
            s.ready(b)
        _SinkLinear.queue.clear()


def _sink_worker(rank, world, port, out, deferred=False):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-mdm_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mdm_hip import distributed as md

    md.init_distributed_singlenode(backend="gloo")
    model = _model(seed=0)
    lin = [m for m in model if isinstance(m, nn.Linear)]
    red = md.GradReducer(list(model.parameters()), bucket_mb=0.002, tail_mb=0.005)
    _SinkLinear.sink = red
    _SinkLinear.deferred = deferred

    def fwd(x):
        for i, m in enumerate(lin):
            x = _SinkLinear.apply(x, m.weight, m.bias)
            if i < len(lin) - 1:
                x = torch.nn.functional.silu(x)
        return x

    g = torch.Generator().manual_seed(100)
    x_all, y_all = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    xs, ys = x_all[rank::world], y_all[rank::world]
    for step in range(2):               # two optimizer steps, each = one no_sync micro-step + one synchronised one
        with red.no_sync():
            ((fwd(xs) - ys) ** 2).mean().backward()
            _SinkLinear.flush()
        ((fwd(xs) - ys) ** 2).mean().backward()
        _SinkLinear.flush()
        red.finish()
        flat = red.flat.clone()
        red.zero_grad()
    if rank == 0:
        torch.save({"flat": flat, "nb": len(red.buckets), "last": red.buckets[-1]}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("deferred", [False, True])
def test_sink_mode_with_accumulation_gloo_world2(tmp_path, deferred):
    """the gradient-sink reporting path (ready() from inside backward, no autograd gradient) through no_sync
    accumulation and several buckets incl. the small tail bucket: every bucket fires exactly once per synchronised
    backward and finish() returns the rank average of the accumulated gradients.  deferred: the gradient is written
    and reported AFTER backward (defer() + ready()); autograd's hook, which fires when the node returns, must not
    release the bucket early."""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_sink_worker, args=(2, _free_port(), out, deferred), nprocs=2, join=True)
    got = torch.load(out)
    model = _model(seed=0)
    g = torch.Generator().manual_seed(100)
    x_all, y_all = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    params = list(model.parameters())
    gs = []
    for r in range(2):
        loss = ((model(x_all[r::2]) - y_all[r::2]) ** 2).mean()
        gs.append(torch.cat([t.reshape(-1) for t in reversed(torch.autograd.grad(loss, params))]))
    expect = (2 * gs[0] + 2 * gs[1]) / 2
    assert torch.allclose(got["flat"], expect, atol=1e-5 * float(expect.abs().max()))
    assert got["nb"] > 2
    s, e = got["last"]
    assert (e - s) == 16 * 64 + 64   # the tail bucket holds exactly the first layer's parameters (<= tail_mb)


def test_single_process_reducer_is_a_noop():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-mdm_amd"))
    from mdm_hip import distributed as md

    model = _model()
    red = md.GradReducer(list(model.parameters()))
    x = torch.randn(4, 16)
    model(x).square().mean().backward()
    red.finish()
    ref = torch.cat([p.grad.reshape(-1) for p in reversed(list(model.parameters()))])
    assert torch.equal(ref, red.flat)
    assert all(p.grad.data_ptr() >= red.flat.data_ptr() for p in model.parameters())
    red.zero_grad()
    assert float(red.flat.abs().max()) == 0.0


def _w4_worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-mdm_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mdm_hip import distributed as md

    md.init_distributed_singlenode(backend="gloo")
    model = _model(seed=rank)
    # bucket sizes that cut the arena at uneven places: a head bucket of one bias-sized parameter (it closes as soon as it
    # is exceeded), 2.2 KB regular buckets that close in the middle of a layer's (weight, bias) pair, the tail bucket = first layer
    red = md.GradReducer(list(model.parameters()), bucket_mb=0.0021, head_mb=0.00002, tail_mb=0.0045)
    assert red.wire_dtype is None          # "auto": no bf16 wire over gloo / on CPU tensors
    red.broadcast_parameters(0)
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(16, 16, generator=g), torch.randn(16, 8, generator=g)
    for step in range(3):
        ((model(x_all[rank::world]) - y_all[rank::world]) ** 2).mean().backward()
        red.finish()
        flat = red.flat.clone()
        red.zero_grad()
    if rank == 0:
        torch.save({"flat": flat, "buckets": red.buckets}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_gloo_world4_uneven_buckets(tmp_path):
    """four ranks, three steps; head / regular / tail buckets of different sizes whose boundaries fall inside and between
    layers: every bucket fires once per step and finish() returns the average over the four ranks"""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_w4_worker, args=(4, _free_port(), out), nprocs=4, join=True)
    got = torch.load(out)
    model = _model(seed=0)
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(16, 16, generator=g), torch.randn(16, 8, generator=g)
    params = list(model.parameters())
    gs = []
    for r in range(4):
        loss = ((model(x_all[r::4]) - y_all[r::4]) ** 2).mean()
        gs.append(torch.cat([t.reshape(-1) for t in reversed(torch.autograd.grad(loss, params))]))
    expect = sum(gs) / 4
    assert torch.allclose(got["flat"], expect, atol=1e-5 * float(expect.abs().max()))
    sizes = [e - s for s, e in got["buckets"]]
    assert len(sizes) >= 4 and sizes[0] == 8                   # the head bucket: the last layer's bias alone
    assert sizes[-1] == 16 * 64 + 64                            # the tail bucket: the first layer
    assert len(set(sizes)) >= 3 and sum(sizes) == expect.numel()


def test_data_parallel_keyword_arguments():
    """DistributedDataParallel keywords: the harmless ones are accepted, find_unused_parameters / static_graph are refused
    with a message (the reducer needs a gradient for every parameter), anything else is a TypeError"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-mdm_amd"))
    from mdm_hip import distributed as md

    m = _model()
    wrapped = md.DataParallel(m, device_ids=[0], output_device=0, broadcast_buffers=False, gradient_as_bucket_view=True,
                              find_unused_parameters=False)
    assert wrapped.module is m and len(wrapped.reducer.buckets) >= 1
    with pytest.raises(ValueError, match="find_unused_parameters"):
        md.DataParallel(_model(), find_unused_parameters=True)
    with pytest.raises(ValueError, match="static_graph"):
        md.DataParallel(_model(), static_graph=True)
    with pytest.raises(TypeError):
        md.DataParallel(_model(), no_such_option=1)


def test_train_batch_refuses_data_parallel_on_the_plain_path():
    """mdm_hip.distributed.DataParallel around a model whose step cannot take the fused path (here: CPU tensors): the plain
    path would neither join the wrapper's all-reduces nor keep p.grad in its arena, so train_batch raises instead of
    letting the ranks drift apart"""
    import types

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-mdm_amd"))
    from mdm_hip import distributed as md
    from mdm_hip import trainer

    class Vision(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(4, 4)

    class Inner(nn.Module):
        def __init__(self):
            super().__init__()
            self.vision_model = Vision()

    class Diffusion(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = md.DataParallel(Inner())

        def get_loss(self, sample):
            raise AssertionError("must not get this far")

    model = Diffusion()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 1.0)
    with pytest.raises(RuntimeError, match="only works with the fused train step"):
        trainer.train_batch(model, {}, opt, sched, None, types.SimpleNamespace(fp16=False))


def test_late_gradient_parameters_go_to_the_end_of_the_arena():
    """the layers a UNet evaluates for the whole network in one grouped launch at the start of forward (every ResNet's
    time_layer, every attention layer's norm_cond / kv_cond) get their gradients when backward ENDS: the reducer puts
    them behind everything else, so that no earlier bucket waits for them (mdm_hip.distributed.GradReducer late_params)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "ml-mdm_amd"), os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import parity_cases as PC
    from mdm_hip import distributed as md

    model, _, _ = PC.build_module("mini_unet")
    late = model.late_gradient_parameters()
    names = {id(p): n for n, p in model.named_parameters()}
    assert late and all(("time_layer" in names[id(p)]) or ("norm_cond" in names[id(p)]) or ("kv_cond" in names[id(p)]) for p in late)
    n_late = sum(p.numel() for p in late)
    red = md.GradReducer(list(model.parameters()), bucket_mb=0.05, head_mb=0.01, tail_mb=0.002, late_params=late, world_override=1)
    total = red.flat.numel()
    late_ids = {id(p) for p in late}
    base = red.flat.data_ptr()
    for p in model.parameters():
        off = (p.grad.data_ptr() - base) // 4
        assert (off >= total - n_late) == (id(p) in late_ids), names[id(p)]
    # the others keep the order backward produces them in (reverse registration)
    early = [p for p in red.order if id(p) not in late_ids]
    assert [id(p) for p in early] == [id(p) for p in reversed(list(model.parameters())) if id(p) not in late_ids]
    assert [id(p) for p in red.order[len(early):]] == [id(p) for p in reversed(list(model.parameters())) if id(p) in late_ids]

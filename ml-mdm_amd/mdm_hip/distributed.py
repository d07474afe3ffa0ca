"""One process per GPU data parallelism over RCCL (``backend="nccl"`` on ROCm) -- the only
parallelism the reference has (clis/train_parallel.py:147-154, distributed.py:27-61).

``init_distributed_singlenode`` keeps the reference's entry point.  ``GradReducer`` replaces
torch DDP's reducer with a design sized for 8 x MI355X on point-to-point xGMI links:

  * the gradients of all parameters live in ONE flat fp32 arena; ``p.grad`` are views into it,
    so there is no bucket copy-in/copy-out and the optimizer tail can stream the arena;
  * the arena is cut into a few LARGE buckets (default 256 MiB, not DDP's 25 MiB): on xGMI a
    collective is per-link bound (7 links x ~153 GB/s per GPU), so fewer, larger all-reduces
    keep every link busy and amortise launch latency; buckets are ordered by the order in which
    backward produces gradients (reverse registration), so the first all-reduce starts while
    most of backward is still running;
  * a bucket's all-reduce is issued from a post-accumulate-grad hook as soon as its last
    gradient is ready, asynchronously: a dedicated COMMUNICATION stream waits for an event recorded
    on every stream that produced gradients of the bucket (main backward chain, weight-gradient side
    stream) and hands the bucket to RCCL; no compute stream ever waits for another one on the
    reducer's account (round 4 made the reporting stream wait for all producers: when that was the
    main stream it stalled behind the whole backlog of the side stream).  ``finish()`` joins before
    the optimizer.  Optional bf16 wire format halves the bytes;
  * the LAST bucket (the first layers of the network: their gradients arrive when backward ends, so
    nothing is left to hide its all-reduce behind) is kept small (``tail_mb``, default 8 MiB): the
    exposed tail of the communication is one sub-millisecond collective, not a 256 MiB one;
  * the FIRST bucket is small too (``head_mb``, default 64 MiB = 4 % of the 64x64 U-Net's 1.66 GB of gradients): the
    links start working within the first tenth of backward instead of after a full 256 MiB has accumulated;
  * the wire format is fp32 by default (``wire_dtype="auto"`` = None = fp32: what the reference's DDP reduces in).
    ``wire_dtype=torch.bfloat16`` halves the bytes per link -- the gradients are divided by the world size in fp32 and
    cast in ONE pass into a persistent wire buffer, the cross-rank SUM then runs in bf16 inside RCCL, the arena the
    optimizer reads stays fp32 -- but it is opt-in: the cast and the copy-back are 4.8 GB of extra HBM traffic per step
    that compete with backward (measured in a world of one: +5 ms per step with 64 MiB buckets, against +2 ms for the
    fp32 wire), while a 1.66 GB fp32 exchange (about 10 ms on a 7-link xGMI ring) hides under a 57 ms backward either
    way; only the exposed tail (the late parameters, about 340 MB) would gain, about a millisecond;
  * ``record_timeline=True`` stamps every bucket's issue and completion with HIP events (``timeline()``), so the first
    run on a multi-GPU node shows how much of the exchange hid behind backward (``bench.py --gpus N`` prints it).

Unmeasured on hardware: the build environment exposes one GPU, so the 1 -> 8 scaling curve is the
driver's to measure; the code path is exercised by gloo world-2 tests on CPU and a 2-rank shared-GPU test.
"""
import os
from datetime import timedelta

import torch
import torch.distributed as dist


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def print0(*a, **k):
    if get_rank() == 0:
        print(*a, **k)


def init_distributed_singlenode(timeout: int = 0, backend: str = None):
    """env:// rendezvous, one rank per GPU (reference distributed.py:27-61).  Returns
    (local_rank, global_rank, world_size); a no-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or "MASTER_ADDR" not in os.environ:
        # single process: still honour LOCAL_RANK / RANK from the environment, as the reference does (:33-40) --
        # a launcher may pin one process per GPU without a rendezvous
        return get_local_rank(), int(os.environ.get("RANK", "0")), 1
    rank, local = int(os.environ["RANK"]), get_local_rank()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {"timeout": timedelta(seconds=timeout)} if timeout else {}
    if backend == "nccl":
        torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank, **kw)
    barrier(local)
    return local, rank, world


def barrier(local_rank: int = None):
    """dist.barrier() that tells RCCL which device this rank uses (without it ProcessGroupNCCL guesses rank % device count
    and warns that a heterogeneous mapping can hang)"""
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[torch.cuda.current_device() if local_rank is None else local_rank])
    else:
        dist.barrier()


class GradReducer:
    """``world_override=1`` keeps the reducer local whatever ``torch.distributed`` holds (an unwrapped model trains on its
    own rank's gradients, like the reference without DDP).  ``force_collectives=True`` issues the bucket all-reduces even
    in a world of one -- the RCCL code path (async all-reduce on arena views, stream ordering, bf16 wire copy-back) can
    then be exercised on a single GPU (``bench.py --force-collectives``)."""

    def __init__(self, params, bucket_mb: float = 256.0, wire_dtype="auto", group=None, tail_mb: float = 8.0,
                 world_override: int = None, force_collectives: bool = False, head_mb: float = 64.0,
                 record_timeline: bool = False, late_params=None):
        """``late_params``: parameters whose gradients are only complete when backward ENDS although they sit in the
        middle of the registration order -- layers the model evaluates for the whole network in one grouped launch at
        the start of forward (``UNet.late_gradient_parameters()``: every ResNet's ``time_layer``, every attention
        layer's text ``norm_cond`` / ``kv_cond``).  They go to the END of the arena: left in place, each of them holds
        its bucket back until the last millisecond of backward (measured: first bucket issued 54.8 ms into a 55.0 ms
        backward, all 27 buckets within the last 3 ms)."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.nranks = dist.get_world_size(group) if dist.is_initialized() else 1   # divisor of the average
        if world_override is not None:
            self.nranks = int(world_override)
        # ``world`` > 1 switches the hooks / buckets / collectives on (a forced single-rank run pretends 2)
        self.world = self.nranks if not (force_collectives and self.nranks == 1 and dist.is_initialized()) else 2
        self._wire_auto = False
        if isinstance(wire_dtype, str):
            if wire_dtype != "auto":
                raise ValueError("wire_dtype: a torch dtype, None (fp32) or 'auto'")
            wire_dtype = None           # "auto" = fp32, the reference's reduction precision
        self.wire_dtype = wire_dtype
        self._wire = {}                 # bucket -> persistent wire buffer (bf16 wire only)
        self._comm = None               # communication stream (GPU tensors, world > 1)
        self._avg_ok = None             # whether the backend takes ReduceOp.AVG (RCCL: yes; decided at the first bucket)
        self.record_timeline = bool(record_timeline)
        self._timeline = []        # (bucket, bytes on the wire, issue event, done event) of the current step
        self._t0 = None
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        # arena order = reverse registration order ~ the order backward produces gradients; the known late ones last
        order = list(reversed(self.params))
        late = {id(p) for p in (late_params or [])}
        if late:
            order = [p for p in order if id(p) not in late] + [p for p in order if id(p) in late]
        self.order = order
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        tail = min(int(tail_mb * (1 << 20) / 4), cap)
        head = max(1, min(int(head_mb * (1 << 20) / 4), cap)) if head_mb else cap
        # index of the first parameter of the tail bucket: the longest suffix of `order` that fits tail_mb
        tail_from, acc = len(order), 0
        for i in range(len(order) - 1, -1, -1):
            acc += order[i].numel()
            if acc > tail:
                break
            tail_from = i
        self.buckets = []          # (start, end)
        self._bucket_of = {}
        off = b_start = 0
        for i, p in enumerate(order):
            if i == tail_from and off > b_start:
                self.buckets.append((b_start, off))
                b_start = off
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self._bucket_of[p] = len(self.buckets)
            off += n
            if off - b_start >= (head if not self.buckets else cap):
                self.buckets.append((b_start, off))
                b_start = off
        if off > b_start:
            self.buckets.append((b_start, off))
        self._slots = {p.data_ptr(): (p, p.grad) for p in self.params}   # gradient-sink lookup (see ops.set_grad_sink)
        self._grad_ptrs = [(p, p.grad.data_ptr()) for p in self.params]
        self._need = [0] * len(self.buckets)
        for p in order:
            self._need[self._bucket_of[p]] += 1
        self._pending = list(self._need)
        self._fired = {}
        self._deferred = set()
        self._streams = [set() for _ in self.buckets]
        self._work = []
        self._enabled = True
        if self.world > 1:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)

    # -- gradient-sink interface (mdm_hip.ops.set_grad_sink): backward kernels add straight into the arena
    def slot(self, p):
        ent = self._slots.get(p.data_ptr())
        return ent[1] if ent is not None and ent[0].shape == p.shape else None

    def ready(self, p):
        if self.world > 1:
            self._count(self._slots[p.data_ptr()][0])

    def defer(self, p):
        """the gradient of ``p`` will be reported by ready() AFTER its autograd node has returned (queued / grouped
        weight-gradient launches, deferred GroupNorm parameter sums).  Autograd's post-accumulate hook fires when the
        node returns -- even though the node handed back no gradient -- and must not count the parameter: the bucket
        would be all-reduced before the gradient is written."""
        if self.world > 1:
            self._deferred.add(id(self._slots[p.data_ptr()][0]))

    def rebind(self):
        """refresh the sink lookup after the parameters' storage moved (e.g. into a flat parameter arena)"""
        self._slots = {p.data_ptr(): (p, p.grad) for p in self.params}

    # -- autograd's post-accumulate hook: the report of every parameter whose gradient autograd itself accumulates
    def _hook(self, p):
        if id(p) in self._deferred:
            return
        self._count(p)

    def note_autocast(self, bf16: bool):
        """kept for callers of the round-4 interface: the wire format no longer follows the step's autocast (see the
        module docstring: bf16 on the wire is an explicit choice)"""
        return None

    def _comm_stream(self):
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=self.flat.device)
        return self._comm

    def _post_stream(self):
        if getattr(self, "_post", None) is None:
            self._post = torch.cuda.Stream(device=self.flat.device)
        return self._post

    # -- once per parameter per synchronised backward (from the hook, or from ready())
    def _count(self, p):
        if not self._enabled:
            return
        b = self._bucket_of[p]
        if id(p) in self._fired:
            # a parameter whose gradient went through the sink reports via ready(); autograd's post-accumulate hook
            # may still fire for it (with an undefined gradient) -- count every parameter once per step
            return
        self._fired[id(p)] = 1
        if p.is_cuda:
            # gradients of one bucket are produced on several HIP streams (main backward chain, weight-gradient side
            # stream): remember them so the all-reduce can be ordered after every producer
            self._streams[b].add(torch.cuda.current_stream())
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        s, e = self.buckets[b]
        buf = self.flat[s:e]
        if not buf.is_cuda:
            buf.div_(self.nranks)  # pre-scale: SUM of pre-divided == AVG, and gloo has no AVG
            if self.wire_dtype is not None and self.wire_dtype != torch.float32:
                wire = buf.to(self.wire_dtype)
                h = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._work.append((h, buf, wire, None))
            else:
                h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._work.append((h, None, None, None))
            return
        # GPU: the communication stream waits for every producer of the bucket (an event at each producer stream's
        # current tail: streams run in order, so that covers its last gradient of the bucket) -- no compute stream is
        # made to wait -- and everything that touches the bucket from here to the copy-back runs on that stream
        comm = self._comm_stream()
        for st in self._streams[b]:
            ev = torch.cuda.Event()
            ev.record(st)
            comm.wait_event(ev)
        self._streams[b] = set()
        post = self._post_stream()
        with torch.cuda.stream(comm):
            ev0 = None
            if self.record_timeline:
                if self._t0 is None:   # first bucket of the step: the clock starts at the first issue
                    self._t0 = torch.cuda.Event(enable_timing=True)
                    self._t0.record(comm)
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record(comm)
            scale = 1.0 / self.nranks   # applied in fp32, whatever goes on the wire
            if self.wire_dtype is not None and self.wire_dtype != torch.float32:
                wire = self._wire.get(b)
                if wire is None:
                    wire = self._wire[b] = torch.empty(e - s, dtype=self.wire_dtype, device=buf.device)
                # ONE pass: fp32 gradient x 1/n -> wire format (4 bytes read + 2 written per element, nothing in place)
                if self.nranks > 1:
                    torch.mul(buf, scale, out=wire)
                else:
                    wire.copy_(buf)
                h = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                nbytes = wire.numel() * wire.element_size()
            else:
                wire = None
                # fp32 on the wire: the 1 / n rides inside the collective (ncclAvg) -- no pass over the bucket at all.  (Round 5
                # scaled every bucket with its own `mul_` on this stream: 1.66 GB read + written per step next to backward.)
                if self._avg_ok is None:
                    self._avg_ok = (dist.get_backend(self.group) == "nccl" and hasattr(dist.ReduceOp, "AVG")
                                    and os.environ.get("MDM_HIP_NO_AVG") != "1")   # (development A/B switch)
                if self._avg_ok and self.nranks > 1:   # (a world of one has nothing to scale; RCCL's AVG of one rank still runs
                    #                                      a kernel over the bucket: +3.9 ms per step, round 6)
                    h = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
                else:
                    if self.nranks > 1:
                        buf.mul_(scale)
                    h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                nbytes = buf.numel() * 4
        # the completion side runs on its own stream, bucket by bucket in issue order: it waits for RCCL (this does not
        # block the host), copies a bf16 wire back into the fp32 arena and stamps the bucket done -- while the
        # communication stream is already preparing the next bucket
        with torch.cuda.stream(post):
            h.wait()
            if wire is not None:
                buf.copy_(wire)
            done = None
            if ev0 is not None:
                done = torch.cuda.Event(enable_timing=True)
                done.record(post)
                self._timeline.append([b, nbytes, ev0, done])
        self._work.append((h, None, None, ev0))

    def no_sync(self):
        """context manager for gradient-accumulation micro-steps (train_parallel.py:201-203)"""
        reducer = self

        class _Ctx:
            def __enter__(self):
                reducer._enabled = False

            def __exit__(self, *a):
                reducer._enabled = True

        return _Ctx()

    def check_bound(self):
        """``p.grad`` must still be the view into the arena it was created as: ``model.zero_grad()`` /
        ``optimizer.zero_grad(set_to_none=True)`` silently detach it, after which gradients would neither be
        all-reduced nor seen by the fused optimizer."""
        for p, ptr in self._grad_ptrs:
            if p.grad is None or p.grad.data_ptr() != ptr:
                raise RuntimeError("a parameter's .grad no longer points into the GradReducer arena (zero_grad(set_to_none=True)?); "
                                   "use GradReducer.zero_grad()")

    def finish(self):
        """join the outstanding all-reduces; afterwards every p.grad holds the rank-average"""
        self.check_bound()
        if self.world > 1 and self._enabled:
            if self._pending == self._need and not self._work:
                self._deferred = set()
                return  # no synchronised backward since the last finish()
            missing = [b for b, n in enumerate(self._pending) if n != 0]
            if missing:
                lost = [(i, tuple(p.shape), self._fired.get(id(p), 0)) for i, p in enumerate(self.params) if self._fired.get(id(p), 0) != 1]
                raise RuntimeError("buckets %s did not receive all their gradients; parameters (index, shape) that "
                                   "did not report exactly once (index, shape, count): %s" % (missing, lost[:12]))
            if self._comm is not None and self.flat.is_cuda:
                # GPU: every bucket's wait + copy-back is already queued on the completion stream; the optimizer's stream
                # waits for that stream ONCE -- the only wait a compute stream does for the reducer
                torch.cuda.current_stream().wait_stream(self._post_stream())
            else:
                for h, buf, wire, _ in self._work:
                    h.wait()
                    if wire is not None:
                        buf.copy_(wire)
            self._work = []
            if self.record_timeline:
                self._last_timeline, self._last_t0 = self._timeline, self._t0
                self._timeline, self._t0 = [], None
            self._pending = list(self._need)
            self._fired = {}
        self._deferred = set()

    def mark(self, what):
        """``record_timeline``: stamp the start ("bw0", before ``loss.backward()``) / the end ("bw1", behind everything
        backward queued on the compute stream, before the join with the communication stream) of a backward pass on
        the current stream, so the bucket times can be read against the window they are meant to hide in"""
        if self.record_timeline and self.flat.is_cuda and self.world > 1:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            setattr(self, "_" + what, ev)

    def backward_window_ms(self):
        """(first bucket issue, end of backward) in ms after the start of backward, of the last finished step"""
        b0, b1, t0 = getattr(self, "_bw0", None), getattr(self, "_bw1", None), getattr(self, "_last_t0", None)
        if b0 is None or b1 is None or t0 is None:
            return None
        torch.cuda.synchronize()
        return b0.elapsed_time(t0), b0.elapsed_time(b1)

    def timeline(self):
        """[(bucket, wire bytes, issue ms, done ms)] of the last finished step, relative to its first issue, both on the
        communication / completion streams: ``issue`` = every producer of the bucket has finished, ``done`` = the averaged
        gradients are back in the arena (record_timeline=True; synchronises the device)"""
        tl, t0 = getattr(self, "_last_timeline", None), getattr(self, "_last_t0", None)
        if not tl or t0 is None:
            return []
        torch.cuda.synchronize()
        return [(b, nb, t0.elapsed_time(e0), t0.elapsed_time(e1) if e1 is not None else float("nan")) for b, nb, e0, e1 in tl]

    def zero_grad(self):
        self.flat.zero_()

    def broadcast_parameters(self, src: int = 0):
        """rank-0 parameters to everyone, as one flat message per 256 MiB (DDP does this at wrap time)"""
        if self.nranks == 1:
            return
        with torch.no_grad():
            chunk, acc = [], 0
            for p in self.params:
                chunk.append(p)
                acc += p.numel()
                if acc * 4 >= (256 << 20):
                    self._bcast(chunk, src)
                    chunk, acc = [], 0
            if chunk:
                self._bcast(chunk, src)

    def _bcast(self, ps, src):
        flat = torch.cat([p.detach().reshape(-1) for p in ps])
        dist.broadcast(flat, src, group=self.group)
        off = 0
        for p in ps:
            p.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()


class DataParallel(torch.nn.Module):
    """Drop-in for ``nn.parallel.DistributedDataParallel(diffusion_model.model, device_ids=[local_rank])``
    (reference clis/train_parallel.py:147-154) that keeps the fused train step: same ``.module`` attribute, same
    ``no_sync()`` context manager (train_parallel.py:201-203), same call-through ``forward``; the gradient exchange is
    a ``GradReducer`` over the vision model's parameters (flat arena, large buckets, tail bucket) instead of DDP's
    25 MiB bucket copies, and rank 0's parameters are broadcast at wrap time like DDP does."""

    # DistributedDataParallel keywords that change nothing here (accepted and ignored) ...
    _IGNORED = {"output_device", "dim", "broadcast_buffers", "process_group", "bucket_cap_mb", "check_reduction",
                "gradient_as_bucket_view", "init_sync", "device_mesh", "mixed_precision", "delay_all_reduce_named_params",
                "param_to_hook_all_reduce"}
    # ... and the ones whose semantics this reducer does not have: every parameter must receive exactly one gradient per
    # synchronised backward (finish() raises otherwise), so an "unused parameters" search or a frozen graph cannot be honoured
    _UNSUPPORTED = {"find_unused_parameters", "static_graph"}

    def __init__(self, module, device_ids=None, bucket_mb: float = 256.0, wire_dtype="auto", tail_mb: float = 8.0,
                 force_collectives: bool = False, head_mb: float = 64.0, record_timeline: bool = False, **ddp_kwargs):
        super().__init__()
        for k, v in ddp_kwargs.items():
            if k in self._UNSUPPORTED:
                if v:
                    raise ValueError("mdm_hip.distributed.DataParallel does not implement %s=True: every parameter of the "
                                     "vision model must receive a gradient in every synchronised backward" % k)
            elif k not in self._IGNORED:
                raise TypeError("DataParallel got an unexpected keyword argument %r" % k)
        if ddp_kwargs.get("process_group") is not None:
            raise ValueError("process_group: pass the group to GradReducer directly (the default group is used here)")
        self.module = module
        net = getattr(module, "vision_model", module)
        late = net.late_gradient_parameters() if hasattr(net, "late_gradient_parameters") else None
        self.reducer = GradReducer(list(net.parameters()), bucket_mb=bucket_mb, wire_dtype=wire_dtype, tail_mb=tail_mb,
                                   force_collectives=force_collectives, head_mb=head_mb, record_timeline=record_timeline,
                                   late_params=late)
        self.reducer.broadcast_parameters(0)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def no_sync(self):
        return self.reducer.no_sync()

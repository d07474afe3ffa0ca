"""torch.autograd bindings of the libmdm_hip C ABI (include/mdm_hip.h).

PyTorch is used here for device memory (tensors as buffers), the current HIP
stream and the autograd tape only; every arithmetic step is a call into the
shared library through raw pointers.  Activations are NHWC tensors
``[N, H, W, C]`` (or ``[rows, C]`` for linear layers) of the compute dtype
(``torch.float32`` = exact parity mode, ``torch.bfloat16`` = throughput mode).
Parameters stay fp32 in the reference's layout; packed copies for the kernels
are cached per parameter version.
"""
import ctypes
import os
import weakref

import torch

from . import _lib

F32, BF16, F32_SPLIT, F32_SPLIT_W = 0, 1, 2, 3

# fp32 tensors, products as three bf16 MFMAs on bf16 hi + lo halves (include/mdm_hip.h MDM_F32_SPLIT): the arithmetic of
# sampling at the reference's precision (it samples in fp32, diffusion.py:181-197) at about a third of the bf16 rate.  Only
# the forward matmul-class launches have the path (convolutions / linears, attention); training in fp32 stays exact.
_fp32_split = False
_weight_planes_on = os.environ.get("MDM_HIP_NO_WEIGHT_PLANES", "0") != "1"   # development A/B: split the weights in the k-loop too


class fp32_split:
    """``with ops.fp32_split():`` (or ``ops.fp32_split(True)`` / ``(False)`` as a switch) -- every matmul-class launch on
    fp32 tensors inside (convolutions, linears, the attention forward; ALSO the input-gradient GEMMs of a backward pass run
    inside the context: they share the forward kernels) forms its products as bf16x3.  No effect on bf16 tensors, on the
    weight-gradient and attention-backward kernels.  It is a sampling mode: a process-wide switch (not thread-safe), read at
    launch time -- a hipGraph captured by ``GraphedSampler`` keeps the arithmetic it was captured with, and the sampler
    keys its graphs on the switch."""

    def __init__(self, enabled: bool = True):
        global _fp32_split
        self._prev = _fp32_split
        _fp32_split = bool(enabled)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        global _fp32_split
        _fp32_split = self._prev


def fp32_split_enabled() -> bool:
    return _fp32_split


def _dt_mm(t: torch.Tensor) -> int:
    """dtype code of a forward matmul-class launch"""
    d = _dt(t)
    return F32_SPLIT if (d == F32 and _fp32_split) else d


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise _lib.MdmHipError("unsupported activation dtype %s" % t.dtype)


def _epv(t: torch.Tensor) -> int:
    return 4 if t.dtype == torch.float32 else 8


def _p(t):
    return None if t is None else t.data_ptr()


# The current stream's raw handle.  torch.cuda.current_stream() builds a Stream object through four Python layers (device
# index lookups, is_available(), an environment read): ~5 us a call, ~1100 calls per nested-256 train step -- a sixth of the
# host's time per step in a profile (tools/host_profile.py), on a step whose small inner-U-Net kernels leave the host little
# lead.  The C entry point behind it returns the same handle in ~0.2 us.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)
if os.environ.get("MDM_HIP_SLOW_STREAM_LOOKUP", "0") == "1":   # development A/B
    _raw_stream = None


def _stream():
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


_stream_objs = {}   # raw handle -> torch Stream object of a stream that has been current (torch never destroys its streams)


def _current_stream_obj():
    raw = _stream()
    st = _stream_objs.get(raw)
    if st is None:
        if len(_stream_objs) > 64:
            _stream_objs.clear()
        st = _stream_objs[raw] = torch.cuda.current_stream()
    return st


def _require_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise _lib.MdmHipError(
            "mdm_hip ops run only on an MI355X (got a %s tensor); there is no CPU fallback" % t.device
        )


def _f32_ws(nbytes: int, device):
    return torch.empty((max(int(nbytes), 4) + 3) // 4, dtype=torch.float32, device=device)


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------------------
# packed-weight cache
# --------------------------------------------------------------------------------------
_wcache = {}  # id(param) -> (weakref to the param, {dtype: (version key, packed tensors)})
_pack_epoch = 0


def invalidate_packed_weights():
    """Force a re-pack of every cached kernel-layout weight on next use.  Needed after parameter updates that
    do not bump ``Tensor._version`` (torch's fused optimizers, our own fused optimizer kernel)."""
    global _pack_epoch
    _pack_epoch += 1


# --------------------------------------------------------------------------------------
# gradient sink: parameter gradients written straight into a flat arena
# --------------------------------------------------------------------------------------
_grad_sink = None


def set_grad_sink(sink):
    """``sink.slot(param)`` -> fp32 view of the parameter's slot in a flat gradient arena (or None) and
    ``sink.ready(param)`` is called once the gradient has been added there.  With a sink installed the backward
    kernels accumulate into the arena (C ABI ``accumulate`` flag) and return no gradient to autograd, which removes
    one ``grad += new`` kernel per parameter per step.  ``None`` restores plain autograd semantics."""
    global _grad_sink
    if _grad_sink is not None and sink is not _grad_sink:
        flush_wgrad_queue()   # queued work refers to the old sink's slots
    _grad_sink = sink


def _slot(param):
    if _grad_sink is None or param is None:
        return None
    return _grad_sink.slot(param)


def _sink_defer(param):
    """tell the sink that ready(param) will come after the autograd node returns (optional part of the protocol)"""
    f = getattr(_grad_sink, "defer", None)
    if f is not None and param is not None:
        f(param)


# Weight gradients are off the critical path of backward (nothing downstream reads them before the optimizer), so
# with a gradient sink installed they can run on a second HIP stream and fill the CUs that the many small kernels
# of the main chain (norm statistics, layer-norm, reductions, tiny GEMMs) leave idle.
_side_stream = None
_async_wgrad = False


def enable_async_wgrad(flag: bool):
    global _async_wgrad
    _async_wgrad = bool(flag)


def side_stream():
    """the weight-gradient stream.  MDM_HIP_SIDE_PRIO (development A/B): its HIP stream priority (-1 high, 0 the default
    streams', 1 low where the runtime has a third level)"""
    global _side_stream
    if _side_stream is None:
        prio = int(os.environ.get("MDM_HIP_SIDE_PRIO", "0"))
        _side_stream = torch.cuda.Stream(priority=prio) if prio else torch.cuda.Stream()
    return _side_stream


# Tensors the side stream is still reading.  Holding a reference does two things: the caching allocator cannot hand the
# block to someone else, and -- the subtle one -- autograd cannot accumulate INTO the tensor in place: a ConvFn /
# FFNFn backward hands `dy` on as the gradient of its residual input, and the engine's InputBuffer adds a second
# incoming gradient in place (on the main stream) only when it holds the sole reference to the first.  Entries are
# dropped once the side-stream event recorded behind their reader has completed, and all at once at the join.
_side_keep = []


def join_side_stream():
    """make the current stream wait for everything queued on the weight-gradient stream"""
    if _side_stream is not None:
        torch.cuda.current_stream().wait_stream(_side_stream)
    _side_keep.clear()   # every later main-stream write is ordered behind the side stream's reads now


_side_record_stream = os.environ.get("MDM_HIP_RECORD_STREAM", "0") == "1"   # development A/B: the old belt-and-braces form
_late_wgrad_first = os.environ.get("MDM_HIP_LATE_WGRAD_FIRST", "1") != "0"     # development A/B (SharedInputLinearsFn.backward)


def _off_critical_path(tensors, fn):
    """run fn() (kernel launches that only write gradient-arena slots) on the side stream when enabled"""
    if not (_async_wgrad and _grad_sink is not None):
        return fn()
    side = side_stream()
    if _raw_stream is None:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
            done = torch.cuda.Event()
            done.record(side)
    else:
        # the same, without the Python layers of current_stream() / the StreamContext (this runs ~120 times per step)
        cur = _current_stream_obj()
        side.wait_stream(cur)
        torch.cuda.set_stream(side)
        try:
            fn()
            done = torch.cuda.Event()
            done.record(side)
        finally:
            torch.cuda.set_stream(cur)
    # The operands stay referenced here until the side stream has passed `done` (the purge below) or the main stream has been
    # made to wait for the side stream (join_side_stream): their blocks cannot be handed out again before that, so no
    # Tensor.record_stream is needed.  (Rounds 2-6 called it as well: the caching allocator then records one event per block
    # on the side stream when the block is freed -- a purge of 64 entries freed ~330 blocks at once, 0.5-1.3 ms of
    # hipEventRecord calls = as many marker packets in the side stream's queue, and at the end of backward, where the main
    # stream waits for the side stream, a gap of that length with the whole GPU idle: tools/fwd_gaps.py --gaps,
    # tools/calls/r6/c39_hip_trace.sh; the host itself is 57-117 ms ahead, profiles/r06_host_lead.txt.)
    live = [t for t in tensors if t is not None]
    if _side_record_stream:
        for t in live:
            t.record_stream(side)
    _side_keep.append((done, live))
    if len(_side_keep) >= 64:
        _side_keep[:] = [e for e in _side_keep if not e[0].query()]


# --------------------------------------------------------------------------------------
# deferred, grouped weight gradients of same-shape 1x1 convolutions
# --------------------------------------------------------------------------------------
# With the gradient sink installed nothing downstream reads a weight gradient before the optimizer, so the launch can
# wait: the (x, dy) pairs of same-shape 1x1 convolutions are queued and, at a flush point (a resolution-level boundary
# of the U-Net, or the end of backward), each queue that fills the chip as ONE grouped launch goes out as
# mdm_conv_wgrad_grouped -- no split of the pixel reduction, no fp32 slabs, no reduce kernel; the others are launched
# one by one as before.  Costs memory (x and dy stay alive until the flush: ~10 GB for UNet-64 at batch 64), not time.
_defer_wgrad = False
_wgrad_queue = {}   # (M, Cin, Cout, device) -> [(x2d, dy2d, weight, bias, slot, bslot)]
_WG_MAX = 32


def enable_deferred_wgrad(flag: bool, max_group: int = 32):
    """``max_group``: layers per grouped launch (<= 32).  Smaller groups hand gradients to the all-reduce earlier
    (multi-GPU runs), larger ones fill the chip better."""
    global _defer_wgrad, _WG_MAX
    flush_wgrad_queue()
    _defer_wgrad = bool(flag)
    _WG_MAX = max(1, min(32, int(max_group)))


def wgrad_grouped(xs, dys, dws, dbs=None):
    """dws[g] (Cout, Cin) += dys[g]^T xs[g]; dbs[g] (Cout) += column sums of dys[g] -- one launch (C ABI
    mdm_conv_wgrad_grouped).  xs[g] [M, Cin], dys[g] [M, Cout] bf16."""
    G = len(xs)
    M, cin = xs[0].shape
    cout = dys[0].shape[1]
    for t in list(xs) + list(dys):
        _require_gpu(t)
    xs, dys = [_c(t) for t in xs], [_c(t) for t in dys]
    arr = lambda vals: (ctypes.c_void_p * G)(*vals)
    _lib.check(_lib.lib().mdm_conv_wgrad_grouped(arr([_p(t) for t in xs]), arr([_p(t) for t in dys]), arr([_p(t) for t in dws]),
                                                 arr([_p(t) for t in dbs]) if dbs is not None else None, G, M, cin, cout,
                                                 _dt(xs[0]), _stream()), "mdm_conv_wgrad_grouped")


def _queue_wgrad(x, dy, weight, bias, slot, bslot, M, cin, cout):
    key = (M, cin, cout, x.device.index)
    q = _wgrad_queue.setdefault(key, [])
    q.append((x, dy, weight, bias, slot, bslot))
    _sink_defer(weight)
    if bslot is not None:
        _sink_defer(bias)
    if len(q) == _WG_MAX:
        _flush_key(key)


def _flush_key(key):
    q = _wgrad_queue.pop(key, None)
    if not q:
        return
    M, cin, cout, _ = key
    L = _lib.lib()
    tile = ctypes.c_int(0)
    _lib.check(L.mdm_conv_wgrad_group_plan(M, cout, cin, BF16, len(q), ctypes.byref(tile)), "mdm_conv_wgrad_group_plan")
    tensors = [t for e in q for t in (e[0], e[1])]
    if tile.value and all((e[5] is not None) == (q[0][5] is not None) for e in q):
        G = len(q)
        arr = lambda vals: (ctypes.c_void_p * G)(*vals)
        xs, dys, dws = arr([_p(e[0]) for e in q]), arr([_p(e[1]) for e in q]), arr([_p(e[4]) for e in q])
        dbs = arr([_p(e[5]) for e in q]) if q[0][5] is not None else None

        def go():
            _prof_wrap("conv_wgrad grouped x%d M=%d N=%d K=%d" % (G, M, cout, cin), 2.0 * G * M * cout * cin, lambda: _lib.check(
                L.mdm_conv_wgrad_grouped(xs, dys, dws, dbs, G, M, cin, cout, BF16, _stream()), "mdm_conv_wgrad_grouped"))
            for e in q:
                _grad_sink.ready(e[2])
                if e[5] is not None:
                    _grad_sink.ready(e[3])
    else:
        def go():
            for x, dy, w, b, slot, bslot in q:
                _wgrad_launch(x, dy, M, 1, 1, cin, 1, 1, cout, 1, 1, out=slot, dbias=bslot)
                _grad_sink.ready(w)
                if bslot is not None:
                    _grad_sink.ready(b)

    _off_critical_path(tensors, go)


def flush_wgrad_queue():
    """launch every queued weight gradient (call at resolution-level boundaries and before the optimizer)"""
    for key in list(_wgrad_queue):
        _flush_key(key)
    flush_gn_params()


# --------------------------------------------------------------------------------------
# deferred GroupNorm parameter gradients
# --------------------------------------------------------------------------------------
# dgamma / dbeta of a GroupNorm are sums over the batch of per-sample terms.  Adding them into the gradient slot with
# atomics from inside the backward kernel costs as much as the kernel itself at the 16x16 level (64 samples x 1536
# addresses on two memory channels: 45 us against 24 us); a reduce kernel per layer is ~100 tiny launches per step.
# With the sink installed nothing reads these gradients before the optimizer, so the kernels store per-sample rows
# into a pooled buffer and ONE launch per flush point (mdm_gn_param_reduce_multi) adds the rows of every pending layer
# into their slots -- in a fixed order, so the result is deterministic as well.
_gn_pending = []     # (pg, pb, slot_g, slot_b, N, C, gamma, beta)
_gn_pool = {}        # device index -> [buffer (fp32), offset, high-water mark of the current step]
_gn_tables = {}      # tuple of (pointers, sizes) -> device descriptor table


def _gn_rows(N, C, device):
    """two [N, C] fp32 row blocks out of the pool (the same addresses every step: the descriptor table is reused)"""
    ent = _gn_pool.get(device.index)
    need = 2 * N * C
    if ent is None or ent[1] + need > ent[0].numel():
        # grow: pending entries keep their old buffer alive through the tensors they hold
        size = max(need, 0 if ent is None else 2 * ent[0].numel(), 16 << 20)
        ent = _gn_pool[device.index] = [torch.empty(size, dtype=torch.float32, device=device), 0]
    off = ent[1]
    ent[1] = off + need
    return ent[0][off:off + N * C], ent[0][off + N * C:off + need]


def flush_gn_params():
    """add the pending per-sample GroupNorm parameter-gradient rows into their gradient slots (one launch)"""
    if not _gn_pending:
        return
    dev = _gn_pending[0][0].device
    key = tuple((e[0].data_ptr(), e[1].data_ptr(), e[2].data_ptr(), e[3].data_ptr(), e[4], e[5]) for e in _gn_pending)
    ent = _gn_tables.get(key)
    if ent is None:
        import numpy as np
        rec = np.zeros(len(key), dtype=[("pg", "<u8"), ("pb", "<u8"), ("dg", "<u8"), ("db", "<u8"), ("N", "<i4"),
                                         ("C", "<i4"), ("first", "<i4"), ("pad", "<i4")])
        blocks = 0
        for i, (pg, pb, dg, db, N, C) in enumerate(key):
            rec[i] = (pg, pb, dg, db, N, C, blocks, 0)
            blocks += (C + 255) // 256
        table = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        if len(_gn_tables) > 64:
            _gn_tables.clear()
        ent = _gn_tables[key] = (table, blocks)
    table, blocks = ent
    _lib.check(_lib.lib().mdm_gn_param_reduce_multi(_p(table), len(key), blocks, _stream()), "mdm_gn_param_reduce_multi")
    for e in _gn_pending:
        _grad_sink.ready(e[6])
        _grad_sink.ready(e[7])
    _gn_pending.clear()
    for ent in _gn_pool.values():
        ent[1] = 0   # the next rows are written by kernels queued behind the reduce on the same stream


class _FlushPointFn(torch.autograd.Function):
    """identity whose backward marks a point of the backward pass at which queued weight gradients are launched"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        flush_wgrad_queue()
        return dy


def wgrad_flush_point(x):
    return _FlushPointFn.apply(x) if (_defer_wgrad and x.requires_grad) else x


def _wgrad_into_sink(x, dy, weight, bias, slot, bslot, N, H, W, cin, Ho, Wo, cout, ks, stride):
    """weight (+ bias) gradient of a convolution whose destinations are gradient-arena slots: queued for a grouped
    launch when it qualifies, else launched now off the critical path"""
    M = N * Ho * Wo
    if _defer_wgrad and _async_wgrad and ks == 1 and x.dtype == torch.bfloat16 and M >= 4096:
        _queue_wgrad(x.reshape(M, cin), dy.reshape(M, cout), weight, bias, slot, bslot, M, cin, cout)
        return

    def go():
        _wgrad_launch(x, dy, N, H, W, cin, Ho, Wo, cout, ks, stride, out=slot, dbias=bslot)
        _grad_sink.ready(weight)
        if bslot is not None:
            _grad_sink.ready(bias)

    _off_critical_path((x, dy), go)


def _cache_slot(weight):
    k = id(weight)
    ent = _wcache.get(k)
    if ent is None or ent[0]() is not weight:
        ent = (weakref.ref(weight, lambda _r, k=k: _wcache.pop(k, None)), {})
        _wcache[k] = ent
    return ent[1]


def _round_up(v, m):
    return (v + m - 1) // m * m


def packed_weight(weight: torch.Tensor, bias, dtype: torch.dtype):
    """(w_fwd, w_dgrad, bias_padded, Cin_pad, Cout_pad) for a reference-layout weight
    ``(Cout, Cin, k, k)`` or ``(Cout, Cin)``; refreshed when the parameter version changes."""
    key = dtype
    ent = _cache_slot(weight)
    ver = (weight._version, None if bias is None else bias._version, weight.data_ptr(), _pack_epoch)
    if key in ent and ent[key][0] == ver:
        return ent[key][1]
    _require_gpu(weight)
    L = _lib.lib()
    cout, cin = weight.shape[0], weight.shape[1]
    ks = weight.shape[2] if weight.dim() == 4 else 1
    epv = 4 if dtype == torch.float32 else 8
    cin_pad, cout_pad = _round_up(cin, epv), _round_up(cout, epv)
    taps = ks * ks
    w32 = _c(weight.detach().float())
    need_d = cin_pad == cin
    prev = ent[key][1] if key in ent else None   # re-pack into the same buffers: their addresses stay valid (repack_all table)
    if prev is not None and prev[0].numel() == cout_pad * taps * cin_pad and prev[0].device == weight.device:
        wf, wd = prev[0], prev[1]
    else:
        wf = torch.empty(cout_pad * taps * cin_pad, dtype=dtype, device=weight.device)
        wd = torch.empty(cin * taps * cout_pad, dtype=dtype, device=weight.device) if need_d else None
        if cout_pad != cout:
            wf.zero_()
    # 3x3: channel-block-major reduction order whenever the channel count is a multiple of the k-tile
    bk = 32 if dtype == torch.float32 else 64
    kbf = bk if (ks == 3 and cin_pad % bk == 0) else 0
    kbd = bk if (ks == 3 and cout_pad % bk == 0) else 0
    _lib.check(
        L.mdm_pack_weight(_p(w32), _p(wf), _p(wd), cout, cin, ks, cin_pad, cout_pad, kbf, kbd,
                          F32 if dtype == torch.float32 else BF16, _stream()),
        "mdm_pack_weight",
    )
    _repacked(wf, wd)
    bp = None
    if bias is not None:
        bp = bias.detach().float()
        if cout_pad != cout:
            bp = torch.cat([bp, bp.new_zeros(cout_pad - cout)])
        bp = _c(bp)
    val = (wf, wd, bp, cin_pad, cout_pad, kbf, kbd)
    ent[key] = (ver, val)
    return val


def packed_s2_dgrad_weight(weight: torch.Tensor):
    """The stride-2 3x3 convolution's weight (Cout, Cin, 3, 3) in the layout of mdm_conv_s2_dgrad: bf16
    [(ph, pw, ci)][co / 64][tap j = 2 dh + dw][co % 64], where dx[2b + p] += dy[b + d] * W[kh(p, d)]:
    (p, d) -> kh = (0, 0) -> 1, (1, 0) -> 2, (1, 1) -> 0 (no tap for (0, 1)).  Cached per parameter version."""
    ent = _cache_slot(weight)
    ver = (weight._version, weight.data_ptr(), _pack_epoch)
    if "s2dgrad" in ent and ent["s2dgrad"][0] == ver:
        return ent["s2dgrad"][1]
    cout, cin = weight.shape[0], weight.shape[1]
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    old = ent.get("s2dgrad")
    packed = old[1] if old is not None and old[1].shape == (4 * cin, cout // 64, 4, 64) else \
        torch.empty((4 * cin, cout // 64, 4, 64), dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.lib().mdm_s2dgrad_pack(_p(w), _p(packed), cout, cin, _stream()), "mdm_s2dgrad_pack")
    ent["s2dgrad"] = (ver, packed)
    return packed


_multi_tables = {}   # dtype -> (signature, device table, n, total blocks)


def repack_all(dtype: torch.dtype):
    """Refresh EVERY cached kernel-layout weight of ``dtype`` in one launch (C ABI mdm_pack_weights_multi) -- call right
    after an optimizer step that invalidated them.  Weights the tiled pack cannot express (channel padding, counts not
    a multiple of 32: the stem and the head) stay stale and are re-packed on their next use as before."""
    import numpy as np

    items = []
    for wref, ent in _wcache.values():
        w = wref()
        if w is None or dtype not in ent or not w.is_cuda:
            continue
        ver, val = ent[dtype]
        wf, wd, bp, cin_pad, cout_pad, kbf, kbd = val
        cout, cin = w.shape[0], w.shape[1]
        if cin_pad != cin or cout_pad != cout or cin % 32 or cout % 32 or w.dtype != torch.float32 or not w.is_contiguous():
            continue
        taps = w.shape[2] * w.shape[3] if w.dim() == 4 else 1
        if taps not in (1, 9):
            continue
        items.append((w, ent, ver, val, cout, cin, taps, kbf, kbd))
    if not items:
        return
    sig = tuple((w.data_ptr(), val[0].data_ptr(), 0 if val[1] is None else val[1].data_ptr()) for w, _, _, val, *_ in items)
    tab = _multi_tables.get(dtype)
    if tab is None or tab[0] != sig:
        desc = np.zeros(len(items), dtype=[("w", "u8"), ("wf", "u8"), ("wd", "u8"), ("cout", "i4"), ("cin", "i4"),
                                           ("taps", "i4"), ("kbf", "i4"), ("kbd", "i4"), ("first", "i4")])
        first = 0
        for i, (w, _, _, val, cout, cin, taps, kbf, kbd) in enumerate(items):
            desc[i] = (w.data_ptr(), val[0].data_ptr(), 0 if val[1] is None else val[1].data_ptr(), cout, cin, taps, kbf, kbd, first)
            first += (cout // 32) * (cin // 32)
        assert desc.dtype.itemsize == 48
        dev_tab = torch.from_numpy(desc.view(np.uint8).copy()).to(items[0][0].device)
        tab = (sig, dev_tab, len(items), first)
        _multi_tables[dtype] = tab
    _lib.check(_lib.lib().mdm_pack_weights_multi(_p(tab[1]), tab[2], tab[3], F32 if dtype == torch.float32 else BF16, _stream()),
               "mdm_pack_weights_multi")
    for w, ent, ver, val, *_ in items:
        ent[dtype] = ((w._version, ver[1], w.data_ptr(), _pack_epoch), val)
        _repacked(val[0], val[1])


# --------------------------------------------------------------------------------------
# optional per-launch timing (bench.py roofline leg): HIP events on the launch stream around every GEMM-class launch
# (with its algorithmic FLOPs) and every large streaming launch (with its algorithmic HBM bytes)
# --------------------------------------------------------------------------------------
_prof = None
_prof_shapes = False


def profile_begin(shapes: bool = False):
    """Start recording.  ``shapes``: key the GEMM table by problem shape as well (tools/)."""
    global _prof, _prof_shapes
    _prof, _prof_shapes = [], shapes


def _prof_wrap(name, work, fn, kind="mfma", executed=None):
    """``work``: algorithmic FLOPs (bytes for kind="hbm") the launch is credited with; ``executed``: the FLOPs it actually
    issues when that differs (the sub-pixel resampling convolutions are credited with the dense 3x3 count of the layer
    they replace and execute 16 / 36 of it)"""
    if _prof is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    if kind == "mfma":
        real = _lib.lib().mdm_last_gemm_kernel()   # the kernel that actually ran (name as rocprofv3 prints it)
        if real:
            name = real.decode() + (name[name.index(" M="):] if " M=" in name else "")
    _prof.append((kind, name, work, e0, e1, work if executed is None else executed))
    return r


def profile_end(peak_tflops, peak_gbs=8000.0):
    """-> the ``roofline`` object of bench.py.  The named kernel is the GEMM-class kernel with the LARGEST TOTAL TIME in
    the step; ``gemm_weighted`` is the FLOP-weighted figure over every GEMM-class launch; ``hbm_kernels`` holds the
    streaming kernels as algorithmic GB/s against the HBM peak."""
    global _prof
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    agg, hbm = {}, {}
    for kind, name, work, e0, e1, executed in rec:
        a = (hbm if kind == "hbm" else agg).setdefault(name, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += work
        a[2] += e0.elapsed_time(e1) * 1e-3
        a[3] += executed
    if not agg:
        return None
    table = {k: {"launches": v[0], "alg_tflop": round(v[1] / 1e12, 3), "time_ms": round(v[2] * 1e3, 3),
                 "tflops": round(v[1] / v[2] / 1e12, 1)} for k, v in agg.items()}
    for k, v in agg.items():
        if abs(v[3] - v[1]) > 1e-6 * v[1]:
            # credited with more FLOPs than it issues: "tflops" is algorithmic credit (it may exceed the MFMA peak),
            # "executed_tflops" is what the matrix pipe does
            table[k]["executed_tflops"] = round(v[3] / v[2] / 1e12, 1)
            table[k]["tflops_is"] = "credit for the dense 3x3 layer this sub-pixel kernel replaces (%.2fx the FLOPs it executes)" % (v[1] / v[3])
    dom = max(agg, key=lambda k: agg[k][2])
    n, fl, t = agg[dom][:3]
    ach = fl / t / 1e12
    tot_f, tot_t = sum(v[1] for v in agg.values()), sum(v[2] for v in agg.values())
    return {
        "bound": "mfma", "achieved": round(ach, 1), "peak": peak_tflops, "unit": "TFLOP/s", "frac": round(ach / peak_tflops, 4),
        "traffic": None, "kernel": dom, "dominant_by": "largest total time among the GEMM-class kernels of one step",
        "launches_per_step": n, "avg_launch_ms": round(t / n * 1e3, 4),
        "alg_gflop_per_launch": round(fl / n / 1e9, 2),
        "gemm_weighted": {"alg_tflop": round(tot_f / 1e12, 2), "time_ms": round(tot_t * 1e3, 2),
                          "tflops": round(tot_f / tot_t / 1e12, 1), "frac": round(tot_f / tot_t / 1e12 / peak_tflops, 4)},
        "all_gemm_kernels": table,
        "hbm_kernels": {k: {"launches": v[0], "alg_gb": round(v[1] / 1e9, 3), "time_ms": round(v[2] * 1e3, 3),
                            "gb_per_s": round(v[1] / v[2] / 1e9, 1), "frac_of_peak": round(v[1] / v[2] / 1e9 / peak_gbs, 4)}
                        for k, v in hbm.items()},
    }


# --------------------------------------------------------------------------------------
# raw launches
# --------------------------------------------------------------------------------------
_conv_plans = {}   # (M, Cout, K, dtype) -> (splits, workspace bytes) of mdm_conv_fwd_plan


def _conv_plan(M, Cout, K, dt):
    key = (M, Cout, K, dt)
    ent = _conv_plans.get(key)
    if ent is None:
        sp, wsb = ctypes.c_int(1), ctypes.c_size_t(0)
        _lib.check(_lib.lib().mdm_conv_fwd_plan(M, Cout, K, dt, ctypes.byref(sp), ctypes.byref(wsb)), "mdm_conv_fwd_plan")
        ent = _conv_plans[key] = (sp.value, wsb.value)
    return ent


def _repacked(*packs):
    """note that these kernel-layout weights were (re-)written: planes made from them are stale"""
    for t in packs:
        if t is not None:
            t._mdm_serial = getattr(t, "_mdm_serial", 0) + 1


def _weight_planes(w: torch.Tensor):
    """The packed fp32 weight ``w`` as bf16 hi / lo planes (C ABI mdm_split_weight_planes), cached on the packed tensor
    until it is re-packed: under MDM_F32_SPLIT the weight operand's split leaves the k-loop (97.8 -> 88.9 ms per
    iteration of the 64x64 sampler at batch 64; the products are bit-identical)."""
    ent, serial = getattr(w, "_mdm_planes", None), getattr(w, "_mdm_serial", 0)
    if ent is not None and ent[0] == serial:
        return ent[1]
    planes = torch.empty_like(w)
    _lib.check(_lib.lib().mdm_split_weight_planes(_p(w), _p(planes), w.numel(), _stream()), "mdm_split_weight_planes")
    w._mdm_planes = (serial, planes)
    return planes


def _conv_launch(x, w, bias, res, aux, y, ypre, N, H, W, Cin, Ho, Wo, Cout, ks, stride, transposed, act, kblk=0):
    # problems too small to fill the chip with output tiles (sampling at batch 1-4) run split over the reduction
    splits, wsb = _conv_plan(N * Ho * Wo, Cout, ks * ks * Cin, _dt(x)) if (stride == 1 and not transposed) else (1, 0)
    ws = _f32_ws(wsb, x.device) if splits > 1 else None
    dt = _dt_mm(x)
    if dt == F32_SPLIT and (ks * ks * Cin) % 8 == 0 and w.numel() % 8 == 0 and _weight_planes_on:
        w, dt = _weight_planes(w), F32_SPLIT_W

    def go():
        _lib.check(
            _lib.lib().mdm_conv_fwd_ws(_p(x), _p(w), _p(bias), _p(res), _p(aux), _p(y), _p(ypre), N, H, W, Cin, Ho, Wo, Cout,
                                       ks, stride, transposed, act, kblk, dt, _p(ws), wsb, _stream()),
            "mdm_conv_fwd",
        )

    if _prof is None:
        return go()
    code = _lib.lib().mdm_conv_fwd_tile(N * Ho * Wo, Cout, _dt(x))
    mode = "1x1" if ks == 1 else ("3x3_T2" if transposed else "3x3")
    name = "conv_gemm_kernel<%s,%dx%d,%s>" % ("f32" if x.dtype == torch.float32 else "bf16", code // 1000, code % 1000, mode)
    flops = 2.0 * N * Ho * Wo * Cout * ks * ks * Cin / (4 if transposed else 1)
    if _prof_shapes:
        name += " M=%d N=%d K=%d" % (N * Ho * Wo, Cout, ks * ks * Cin)
    return _prof_wrap(name, flops, go)


def _wgrad_launch(x, dy, N, H, W, Cin, Ho, Wo, Cout, ks, stride, out=None, dbias=None):
    """dW (Cout, Cin, ks, ks) fp32; with ``out`` (a gradient-arena slot) the result is ADDED into it.  ``dbias`` (fp32
    [Cout]) receives the bias gradient from the same launch, with the same overwrite / accumulate semantics."""
    L = _lib.lib()
    splits = ctypes.c_int(0)
    wsb = ctypes.c_size_t(0)
    M, K = N * Ho * Wo, ks * ks * Cin
    _lib.check(L.mdm_conv_wgrad_plan(M, Cout, K, _dt(x), ctypes.byref(splits), ctypes.byref(wsb)), "mdm_conv_wgrad_plan")
    ws = _f32_ws(wsb.value, x.device)
    dw = torch.empty((Cout, Cin, ks, ks), dtype=torch.float32, device=x.device) if out is None else out

    def go():
        _lib.check(L.mdm_conv_wgrad(_p(x), _p(dy), 0 if dbias is None else 1, _p(ws), N, H, W, Cin, Ho, Wo, Cout, ks, stride,
                                    _dt(x), _stream()),
                   "mdm_conv_wgrad")

    if _prof is None:
        go()
    else:
        te = L.mdm_conv_wgrad_tile(M, Cout, K, _dt(x))
        kn = "conv_wgrad_kernel" if x.dtype == torch.float32 else "conv_wgrad_tr_kernel"
        name = "%s<%s,%dx%d,%s>" % (kn, "f32" if x.dtype == torch.float32 else "bf16", te, te, "1x1" if ks == 1 else "3x3")
        if _prof_shapes:
            name += " M=%d N=%d K=%d" % (M, Cout, K)
        _prof_wrap(name, 2.0 * M * Cout * K, go)
    _lib.check(L.mdm_conv_wgrad_reduce(_p(ws), _p(dw), _p(dbias), _p(dy), M, Cin, Cout, ks, 0 if out is None else 1, _dt(x),
                                       _stream()), "mdm_conv_wgrad_reduce")
    return dw


def _colsum_launch(x2d, M, C, out=None):
    L = _lib.lib()
    nb = ctypes.c_int(0)
    wsb = ctypes.c_size_t(0)
    _lib.check(L.mdm_colsum_plan(M, C, ctypes.byref(nb), ctypes.byref(wsb)), "mdm_colsum_plan")
    ws = _f32_ws(wsb.value, x2d.device)
    acc = 0 if out is None else 1
    if out is None:
        out = torch.empty(C, dtype=torch.float32, device=x2d.device)
    _lib.check(L.mdm_colsum(_p(x2d), _p(out), _p(ws), M, C, acc, _dt(x2d), _stream()), "mdm_colsum")
    return out


def _geom(x, ks, stride):
    """x: [N,H,W,C] or [R,C] -> (N,H,W,Ho,Wo)"""
    if x.dim() == 2:
        return x.shape[0], 1, 1, 1, 1
    N, H, W = x.shape[0], x.shape[1], x.shape[2]
    if ks == 1:
        return N, H, W, H, W
    return N, H, W, (H - 1) // stride + 1, (W - 1) // stride + 1


def _out_shape(x, Ho, Wo, C):
    return (x.shape[0], C) if x.dim() == 2 else (x.shape[0], Ho, Wo, C)


class ConvFn(torch.autograd.Function):
    """y = conv(x, weight) + bias (+ residual).  3x3 (stride 1/2, pad 1), 1x1, or linear ([R, Cin])."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, stride, tap=False):
        """``tap``: also return x itself as a second output -- for a consumer that reads the same tensor (the skip
        connection of a down-sampling block): its gradient comes back to ``backward`` and is added in the input
        gradient's epilogue instead of by autograd's accumulation pass"""
        _require_gpu(x)
        x = _c(x)
        residual = _c(residual)
        ks = weight.shape[2] if weight.dim() == 4 else 1
        wf, wd, bp, cin_pad, cout_pad, kbf, kbd = packed_weight(weight, bias, x.dtype)
        if x.shape[-1] != cin_pad:
            raise _lib.MdmHipError("conv input has %d channels, packed weight expects %d" % (x.shape[-1], cin_pad))
        N, H, W, Ho, Wo = _geom(x, ks, stride)
        y = torch.empty(_out_shape(x, Ho, Wo, cout_pad), dtype=x.dtype, device=x.device)
        _conv_launch(x, wf, bp, residual, None, y, None, N, H, W, cin_pad, Ho, Wo, cout_pad, ks, stride, 0, 0, kbf)
        ctx.save_for_backward(x, weight, bias)
        ctx.stride, ctx.ks = stride, ks
        ctx.has_res = residual is not None
        if tap:
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dtap=None):
        x, weight, bias = ctx.saved_tensors
        if dy is None:
            raise _lib.MdmHipError("conv(tap=True): the convolution's own output received no gradient")
        return _conv_backward(ctx, x, weight, bias, dy, dtap) + (None, None)


def _conv_backward(ctx, x, weight, bias, dy, dtap=None):
    """input / weight / bias / residual gradients of y = conv(x, weight) + bias (+ residual); ``ctx`` carries ks, stride,
    has_res and needs_input_grad[0:4] = (x, weight, bias, residual); ``dtap``: a second gradient of x, added to dx.
    -> (dx, dw, db, dres)"""
    dy = _c(dy)
    dtap = _c(dtap)
    ks, stride = ctx.ks, ctx.stride
    wf, wd, bp, cin_pad, cout_pad, kbf, kbd = packed_weight(weight, bias, x.dtype)
    cout, cin = weight.shape[0], weight.shape[1]
    N, H, W, Ho, Wo = _geom(x, ks, stride)
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        if wd is None:
            raise _lib.MdmHipError("input gradient requested for a channel-padded convolution")
        dx = torch.empty_like(x)
        if ks == 3 and stride == 2 and x.dtype == torch.bfloat16 and cout_pad == cout and cout % 64 == 0 and cin % 128 == 0 \
                and H % 2 == 0 and W % 2 == 0:
            # sub-pixel form: a 2x2 correlation over dy per phase of dx, stored pixel-shuffled (mdm_conv_s2_dgrad)
            wsel = packed_s2_dgrad_weight(weight)
            _prof_wrap("conv_gemm_bl_kernel<sel4> (3x3 stride-2 input gradient) M=%d N=%d K=%d" % (N * Ho * Wo, 4 * cin, 4 * cout),
                       2.0 * N * Ho * Wo * cout * 9 * cin, lambda: _lib.check(
                _lib.lib().mdm_conv_s2_dgrad_res(_p(dy), _p(wsel), _p(dtap), _p(dx), N, Ho, Wo, cout, cin, BF16, _stream()),
                "mdm_conv_s2_dgrad_res"), executed=2.0 * N * Ho * Wo * (4 * cin) * (4 * cout))
            dtap = None
        elif ks == 3 and stride == 2:
            _conv_launch(dy, wd, None, None, None, dx, None, N, Ho, Wo, cout_pad, H, W, cin, 3, 1, 1, 0, kbd)
        else:
            _conv_launch(dy, wd, None, None, None, dx, None, N, Ho, Wo, cout_pad, H, W, cin, ks, 1, 0, 0, kbd)
        if dtap is not None:   # shapes the sub-pixel kernel does not take
            dx = dx + dtap
    elif dtap is not None:
        dx = dtap
    padded = cout_pad != cout or cin_pad != cin
    want_b = bias is not None and ctx.needs_input_grad[2]
    if ctx.needs_input_grad[1]:
        slot = None if padded else _slot(weight)
        bslot = _slot(bias) if (slot is not None and want_b) else None
        if slot is not None:
            # weight (and, when it also lives in the arena, bias) gradient from one launch
            _wgrad_into_sink(x, dy, weight, bias, slot, bslot, N, H, W, cin_pad, Ho, Wo, cout_pad, ks, stride)
            if bslot is not None:
                want_b = False
        else:
            dbt = torch.empty(cout_pad, dtype=torch.float32, device=x.device) if want_b else None
            dwp = _wgrad_launch(x, dy, N, H, W, cin_pad, Ho, Wo, cout_pad, ks, stride, dbias=dbt)
            if padded:
                dwp = dwp[:cout, :cin].contiguous()
            dw = dwp.view(weight.shape)
            if want_b:
                db = dbt[:cout]
                want_b = False
    if want_b:
        slot = None if cout_pad != cout else _slot(bias)
        if slot is not None:
            _colsum_launch(dy, N * Ho * Wo, cout_pad, out=slot)
            _grad_sink.ready(bias)
        else:
            db = _colsum_launch(dy, N * Ho * Wo, cout_pad)[:cout]
    dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
    return dx, dw, db, dres


def conv(x, weight, bias=None, residual=None, stride=1):
    return ConvFn.apply(x, weight, bias, residual, stride)


def conv_tap(x, weight, bias=None, stride=1):
    """(conv(x), x'): x' is x for a second consumer whose gradient is then added inside the convolution's input-gradient
    kernel (mdm_conv_s2_dgrad_res for the stride-2 3x3 case) rather than by a separate accumulation pass"""
    if not (torch.is_grad_enabled() and x.requires_grad) or os.environ.get("MDM_HIP_NO_CONV_TAP", "0") == "1":
        return ConvFn.apply(x, weight, bias, None, stride), x
    return ConvFn.apply(x, weight, bias, None, stride, True)


# --------------------------------------------------------------------------------------
# convolution + the GroupNorm that reads its output, one launch
# --------------------------------------------------------------------------------------
def conv_gn_enabled():
    """whether the model uses the fused launch where it applies.  Off unless MDM_HIP_CONV_GN=1: measured in one call
    against conv + group_norm on the 64x64 U-Net step (profiles/r03_did_not_pay.md) it is neutral -- the norm kernel it
    removes is an HBM-bound launch that already overlaps the weight-gradient stream, and the epilogue's three passes over
    the staged tile cost the convolution what the norm cost."""
    return os.environ.get("MDM_HIP_CONV_GN", "0") == "1"


def conv_gn_supported(x, weight, gamma, groups, stride=1):
    """the fused epilogue takes bf16 16x16 images with 24 channels per group (the 768-channel level of the 64x64 U-Net)"""
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.dim() != 4 or stride != 1:
        return False
    ks = weight.shape[2] if weight.dim() == 4 else 1
    cin, cout = weight.shape[1], weight.shape[0]
    kb = 64 if (ks == 3 and cin % 64 == 0) else 0
    return bool(_lib.lib().mdm_conv_fwd_gn_ok(x.shape[0], x.shape[1], x.shape[2], cin, cout, ks, kb, groups, BF16)) and \
        gamma.numel() == cout


class ConvGNFn(torch.autograd.Function):
    """(y, y_norm) with y = conv(x, weight) + bias (+ residual) and y_norm = act(GroupNorm(y; gamma, beta)): the norm's
    statistics and its normalised output come out of the convolution's epilogue (C ABI mdm_conv_fwd_gn) -- the separate
    norm kernel and its read of y are gone.  Backward = the GroupNorm backward kernel (the gradient that reaches y
    directly -- the residual branch, a skip connection -- rides in as its ``dres``) followed by the convolution's."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, gamma, beta, groups, eps, act):
        _require_gpu(x)
        x, residual = _c(x), _c(residual)
        ks = weight.shape[2] if weight.dim() == 4 else 1
        wf, wd, bp, cin_pad, cout_pad, kbf, kbd = packed_weight(weight, bias, x.dtype)
        N, H, W = x.shape[0], x.shape[1], x.shape[2]
        g32, b32 = _c(gamma.detach().float()), _c(beta.detach().float())
        y = torch.empty((N, H, W, cout_pad), dtype=x.dtype, device=x.device)
        yn = torch.empty_like(y)
        stats = torch.empty((N, groups, 2), dtype=torch.float32, device=x.device)
        coef = torch.empty((N, cout_pad, 2), dtype=torch.float32, device=x.device)
        _prof_wrap("conv_gemm_bl_kernel<256, 192, +gn> M=%d N=%d K=%d" % (N * H * W, cout_pad, ks * ks * cin_pad),
                   2.0 * N * H * W * cout_pad * ks * ks * cin_pad, lambda: _lib.check(
            _lib.lib().mdm_conv_fwd_gn(_p(x), _p(wf), _p(bp), _p(residual), _p(y), N, H, W, cin_pad, cout_pad, ks, kbf, _p(g32), _p(b32),
                                       groups, float(eps), act, _p(yn), _p(stats), _p(coef), BF16, _stream()), "mdm_conv_fwd_gn"))
        ctx.save_for_backward(x, weight, bias, y, gamma, beta, stats, coef)
        ctx.ks, ctx.stride, ctx.has_res = ks, 1, residual is not None
        ctx.groups, ctx.act = groups, act
        return y, yn

    @staticmethod
    def backward(ctx, dy_direct, dyn):
        x, weight, bias, y, gamma, beta, stats, coef = ctx.saved_tensors
        # gradient w.r.t. y: through the norm (dyn) plus whatever reached y directly
        dy, dgamma, dbeta, _ = _gn_backward(dyn, y, gamma, beta, None, stats, coef, dy_direct, None, ctx.groups, ctx.act)
        dx, dw, db, dres = _conv_backward(ctx, x, weight, bias, dy)
        return dx, dw, db, dres, dgamma, dbeta, None, None, None


def conv_gn(x, weight, bias, residual, gamma, beta, groups, eps=1e-5, silu=False):
    """-> (y, act(GroupNorm(y))) with y = conv(x) + bias (+ residual); see ConvGNFn / conv_gn_supported"""
    return ConvGNFn.apply(x, weight, bias, residual, gamma, beta, groups, eps, 1 if silu else 0)


# --------------------------------------------------------------------------------------
# upsample2x (nearest) -> conv3x3 in its sub-pixel form
# --------------------------------------------------------------------------------------
def packed_upconv_weights(weight: torch.Tensor, bias):
    """(w_ph, w_t, bias4) of mdm_conv_up_fwd / mdm_conv_up_dgrad for a (Cout, Cin, 3, 3) weight; cached per version."""
    ent = _cache_slot(weight)
    ver = (weight._version, None if bias is None else bias._version, weight.data_ptr(), _pack_epoch)
    if "upconv" in ent and ent["upconv"][0] == ver:
        return ent["upconv"][1]
    cout, cin = weight.shape[0], weight.shape[1]
    prev = ent["upconv"][1] if "upconv" in ent else None
    if prev is not None and prev[0].device == weight.device:
        w_ph, w_t = prev[0], prev[1]
    else:
        w_ph = torch.empty(4 * cout * 4 * cin, dtype=torch.bfloat16, device=weight.device)
        w_t = torch.empty(cin * 16 * cout, dtype=torch.bfloat16, device=weight.device)
    _lib.check(_lib.lib().mdm_upconv_pack(_p(_c(weight.detach().float())), _p(w_ph), _p(w_t), cout, cin, _stream()), "mdm_upconv_pack")
    bias4 = _c(bias.detach().float().repeat(4)) if bias is not None else None
    val = (w_ph, w_t, bias4)
    ent["upconv"] = (ver, val)
    return val


def upsample_conv_supported(x, weight):
    """bf16, channel counts the 256-square weight-gradient tiles divide, power-of-two images (the sampling / fp32 /
    odd-size cases take upsample2x + conv)"""
    if os.environ.get("MDM_HIP_UPCONV", "1") == "0":   # development A/B switch
        return False
    return x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and weight.dim() == 4 and weight.shape[2] == 3 and \
        weight.shape[0] % 256 == 0 and weight.shape[1] % 256 == 0 and x.shape[-1] == weight.shape[1] and \
        (x.shape[1] & (x.shape[1] - 1)) == 0 and (x.shape[2] & (x.shape[2] - 1)) == 0


class UpsampleConvFn(torch.autograd.Function):
    """y = conv3x3(upsample2x_nearest(x), weight) + bias (reference unet.py:567-569) without the upsampled tensor: per
    output phase a 2x2 correlation over the low-resolution x with the 3x3 taps that coincide summed -- 16 weight blocks
    instead of 36 in all three directions (C ABI mdm_conv_up_fwd / _dgrad / mdm_conv_wgrad_blocked + mdm_upconv_wfold)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_gpu(x)
        x = _c(x)
        N, H, W, cin = x.shape
        cout = weight.shape[0]
        w_ph, _, bias4 = packed_upconv_weights(weight, bias)
        y = torch.empty((N, 2 * H, 2 * W, cout), dtype=x.dtype, device=x.device)
        _prof_wrap("conv_gemm_bl_kernel<sel4> (upsample2x + 3x3) M=%d N=%d K=%d" % (N * H * W, 4 * cout, 4 * cin),
                   2.0 * N * 4 * H * W * cout * 9 * cin, lambda: _lib.check(
            _lib.lib().mdm_conv_up_fwd(_p(x), _p(w_ph), _p(bias4), _p(y), N, H, W, cin, cout, BF16, _stream()), "mdm_conv_up_fwd"),
                   executed=2.0 * N * H * W * (4 * cout) * (4 * cin))
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        dy = _c(dy)
        N, H, W, cin = x.shape
        cout = weight.shape[0]
        L = _lib.lib()
        _, w_t, _ = packed_upconv_weights(weight, bias)
        dyb = torch.empty((N, H, W, 4 * cout), dtype=dy.dtype, device=dy.device)
        _lib.check(L.mdm_space_to_depth2x(_p(dy), _p(dyb), N, H, W, cout, BF16, _stream()), "mdm_space_to_depth2x")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _prof_wrap("conv_gemm_bl_kernel<sel4> (upsample2x + 3x3 input gradient) M=%d N=%d K=%d" % (N * H * W, cin, 16 * cout),
                       2.0 * N * 4 * H * W * cout * 9 * cin, lambda: _lib.check(
                L.mdm_conv_up_dgrad(_p(dyb), _p(w_t), _p(dx), N, H, W, cout, cin, BF16, _stream()), "mdm_conv_up_dgrad"),
                           executed=2.0 * N * H * W * cin * (16 * cout))
        if ctx.needs_input_grad[1]:
            want_b = bias is not None and ctx.needs_input_grad[2]
            slot = _slot(weight)
            bslot = _slot(bias) if (slot is not None and want_b) else None
            sunk = slot is not None and (bslot is not None or not want_b)
            M, K = N * H * W, 9 * cin
            splits, wsb = ctypes.c_int(0), ctypes.c_size_t(0)
            _lib.check(L.mdm_conv_wgrad_plan(M, 4 * cout, K, BF16, ctypes.byref(splits), ctypes.byref(wsb)), "mdm_conv_wgrad_plan")

            def go():
                ws = _f32_ws(wsb.value, x.device)
                dwb = torch.empty((4 * cout, cin, 3, 3), dtype=torch.float32, device=x.device)
                db4 = torch.empty(4 * cout, dtype=torch.float32, device=x.device) if want_b else None
                _prof_wrap("conv_wgrad_bl_kernel<1, 1> (upsample2x + 3x3, 16 of 36 blocks) M=%d N=%d K=%d" % (M, 4 * cout, K),
                           2.0 * 4 * M * cout * K, lambda: _lib.check(
                    L.mdm_conv_wgrad_blocked(_p(x), _p(dyb), 1 if want_b else 0, _p(ws), N, H, W, cin, cout, BF16, _stream()),
                    "mdm_conv_wgrad_blocked"), executed=2.0 * 4 * M * cout * K * 16.0 / 36.0)
                _lib.check(L.mdm_conv_wgrad_reduce(_p(ws), _p(dwb), _p(db4), _p(dyb), M, cin, 4 * cout, 3, 0, BF16, _stream()),
                           "mdm_conv_wgrad_reduce")
                dwo = slot if sunk else torch.empty(weight.shape, dtype=torch.float32, device=x.device)
                dbo = (bslot if sunk else torch.empty(cout, dtype=torch.float32, device=x.device)) if want_b else None
                _lib.check(L.mdm_upconv_wfold(_p(dwb), _p(dwo), _p(db4), _p(dbo), cout, cin, 1 if sunk else 0, _stream()), "mdm_upconv_wfold")
                if sunk:
                    _grad_sink.ready(weight)
                    if want_b:
                        _grad_sink.ready(bias)
                return dwo, dbo

            if sunk:
                _off_critical_path((x, dyb), go)
            else:
                dw, db = go()
        return dx, dw, db


def upsample_conv(x, weight, bias=None):
    """conv3x3(upsample2x(x)): the sub-pixel form where it applies, else the two separate ops"""
    if upsample_conv_supported(x, weight):
        return UpsampleConvFn.apply(x, weight, bias)
    return ConvFn.apply(upsample2x(x), weight, bias, None, 1)


def linear(x, weight, bias=None, residual=None):
    shp = x.shape
    y = ConvFn.apply(x.reshape(-1, shp[-1]), weight, bias, None if residual is None else residual.reshape(-1, weight.shape[0]), 1)
    return y.reshape(*shp[:-1], weight.shape[0])


class FFNFn(torch.autograd.Function):
    """y = W2 gelu(W1 x + b1) + b2 + residual  (1x1 convs; unet.py:266-272, 311-312)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, keep):
        _require_gpu(x)
        x, residual = _c(x), _c(residual)
        wf1, _, bp1, c_in, c_hid, _, _ = packed_weight(w1, b1, x.dtype)
        wf2, _, bp2, _, c_out, _, _ = packed_weight(w2, b2, x.dtype)
        N, H, W, _, _ = _geom(x, 1, 1)
        # what backward needs of the pre-activation: itself for fp32 tensors; for bf16 tensors one byte per element, the
        # code of gelu'(pre) (include/mdm_hip.h, MDM_ACT_GELU) -- half the bytes of this store-bound launch's second output
        byte_code = x.dtype == torch.bfloat16 and _lib.lib().mdm_dev_ffn_aux_bytes() == 1
        pre = torch.empty(_out_shape(x, H, W, c_hid), dtype=torch.uint8 if byte_code else x.dtype, device=x.device) if keep else None
        a = torch.empty(_out_shape(x, H, W, c_hid), dtype=x.dtype, device=x.device)
        _conv_launch(x, wf1, bp1, None, None, a, pre, N, H, W, c_in, H, W, c_hid, 1, 1, 0, 1)
        y = torch.empty(_out_shape(x, H, W, c_out), dtype=x.dtype, device=x.device)
        _conv_launch(a, wf2, bp2, residual, None, y, None, N, H, W, c_hid, H, W, c_out, 1, 1, 0, 0)
        ctx.save_for_backward(x, w1, b1, w2, b2, pre, a)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2, b2, pre, a = ctx.saved_tensors
        dy = _c(dy)
        _, wd1, _, c_in, c_hid, _, _ = packed_weight(w1, b1, x.dtype)
        _, wd2, _, _, c_out, _, _ = packed_weight(w2, b2, x.dtype)
        N, H, W, _, _ = _geom(x, 1, 1)
        M = N * H * W
        dpre = torch.empty_like(a)
        _conv_launch(dy, wd2, None, None, pre, dpre, None, N, H, W, c_out, H, W, c_hid, 1, 1, 0, 2)
        def wb_grads(xin, g, w, b, cin_, cout_):
            """(dW, db) of a 1x1 conv from ONE wgrad launch; None entries went into the gradient arena"""
            sw, sb = _slot(w), _slot(b)
            if sw is not None and sb is not None:
                _wgrad_into_sink(xin, g, w, b, sw, sb, N, H, W, cin_, H, W, cout_, 1, 1)
                return None, None
            dbt = torch.empty(cout_, dtype=torch.float32, device=xin.device)
            return _wgrad_launch(xin, g, N, H, W, cin_, H, W, cout_, 1, 1, dbias=dbt).view(w.shape), dbt

        dw2, db2 = wb_grads(a, dy, w2, b2, c_hid, c_out)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _conv_launch(dpre, wd1, None, None, None, dx, None, N, H, W, c_hid, H, W, c_in, 1, 1, 0, 0)
        dw1, db1 = wb_grads(x, dpre, w1, b1, c_in, c_hid)
        dres = dy if (ctx.has_res and ctx.needs_input_grad[5]) else None
        return dx, dw1, db1, dw2, db2, dres, None


def ffn(x, w1, b1, w2, b2, residual):
    # autograd.Function.forward runs with grad mode off, so the "will there be a backward" decision is taken here
    return FFNFn.apply(x, w1, b1, w2, b2, residual, torch.is_grad_enabled())


# --------------------------------------------------------------------------------------
# normalisation
# --------------------------------------------------------------------------------------
def _gn_ws(N, HW, C, G, device):
    wsb = ctypes.c_size_t(0)
    _lib.check(_lib.lib().mdm_gn_plan(N, HW, C, G, ctypes.byref(wsb)), "mdm_gn_plan")
    return _f32_ws(wsb.value, device)


class GroupNormFn(torch.autograd.Function):
    """y = act(GroupNorm(x) * (1 + film[:, :C]) + film[:, C:]);  act in {0: none, 1: SiLU}."""

    @staticmethod
    def forward(ctx, x, gamma, beta, film, groups, eps, act, passthrough):
        _require_gpu(x)
        x, film = _c(x), _c(film)
        N, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (N * C)
        g32, b32 = _c(gamma.detach().float()), _c(beta.detach().float())
        y = torch.empty_like(x)
        stats = torch.empty((N, groups, 2), dtype=torch.float32, device=x.device)
        coef = torch.empty((N, C, 2), dtype=torch.float32, device=x.device)
        ws = _gn_ws(N, HW, C, groups, x.device)
        _prof_wrap("group_norm fwd (HW=%d)" % HW, 2.0 * x.numel() * x.element_size(), lambda: _lib.check(
            _lib.lib().mdm_gn_fwd(_p(x), _p(g32), _p(b32), _p(film), _p(y), _p(stats), _p(coef), _p(ws), N, HW, C,
                                  groups, float(eps), act, _dt(x), _stream()),
            "mdm_gn_fwd",
        ), kind="hbm")
        ctx.save_for_backward(x, gamma, beta, film, stats, coef)
        ctx.groups, ctx.act = groups, act
        ctx.passthrough = passthrough
        # an unused pass-through output (the skip tap of a block whose activations nobody collects) must reach backward
        # as None, not as a materialised zero tensor that costs an allocation, a memset and a read pass of the kernel
        ctx.set_materialize_grads(False)
        if passthrough == 2:
            # third output = x once more, for a consumer OUTSIDE the block (the U-Net's skip connection): its gradient
            # arrives as dres2 and is added in the same kernel -- no accumulation kernel of the autograd engine
            return y, x.view_as(x), x.view_as(x)
        if passthrough:
            # second output = x itself: the caller routes the block's residual branch through it, so the gradient of
            # that branch arrives HERE (dres) and is added inside the GroupNorm backward kernel
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dres=None, dres2=None):
        x, gamma, beta, film, stats, coef = ctx.saved_tensors
        if dy is None:                       # only the pass-through outputs were used
            dy = torch.zeros_like(x)
        dx, dgamma, dbeta, dfilm = _gn_backward(dy, x, gamma, beta, film, stats, coef, dres, dres2, ctx.groups, ctx.act)
        return dx, dgamma, dbeta, dfilm, None, None, None, None


def _gn_backward(dy, x, gamma, beta, film, stats, coef, dres, dres2, groups, act):
    """the GroupNorm backward launch (+ where its parameter gradients go): -> (dx, dgamma, dbeta, dfilm); dgamma / dbeta
    are None when they went into the gradient sink (now, or as per-sample rows reduced at the next flush point)"""
    dy = _c(dy)
    dres = _c(dres) if dres is not None else None
    dres2 = _c(dres2) if dres2 is not None else None
    if dres is None and dres2 is not None:
        dres, dres2 = dres2, None
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    g32, b32 = _c(gamma.detach().float()), _c(beta.detach().float())
    dx = torch.empty_like(x)
    sg, sb = _slot(gamma), _slot(beta)
    sunk = sg is not None and sb is not None
    deferred = sunk and _defer_wgrad   # per-sample rows now, one multi-layer reduce at the next flush point
    if deferred:
        dgamma, dbeta = _gn_rows(N, C, x.device)
    else:
        dgamma = sg if sunk else torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = sb if sunk else torch.empty(C, dtype=torch.float32, device=x.device)
    dfilm = torch.empty_like(film) if film is not None else None
    ws = _gn_ws(N, HW, C, groups, x.device)
    mode = 2 if deferred else (1 if sunk else 0)
    npass = 3.0 + (dres is not None) + (dres2 is not None)
    _prof_wrap("group_norm bwd (HW=%d)" % HW, npass * x.numel() * x.element_size(), lambda: _lib.check(
        _lib.lib().mdm_gn_bwd(_p(dy), _p(x), _p(g32), _p(b32), _p(film), _p(stats), _p(coef), _p(dres), _p(dres2), _p(dx), _p(dgamma),
                              _p(dbeta), _p(dfilm), _p(ws), N, HW, C, groups, act, mode, _dt(x), _stream()),
        "mdm_gn_bwd",
    ), kind="hbm")
    if deferred:
        _gn_pending.append((dgamma, dbeta, sg, sb, N, C, gamma, beta))
        _sink_defer(gamma)
        _sink_defer(beta)
        return dx, None, None, dfilm
    if sunk:
        _grad_sink.ready(gamma)
        _grad_sink.ready(beta)
        return dx, None, None, dfilm
    return dx, dgamma, dbeta, dfilm


def group_norm(x, gamma, beta, groups, eps=1e-5, film=None, silu=False, passthrough=False):
    """``passthrough=True`` returns ``(y, x_res)``: use ``x_res`` (== x) for the residual branch of the block this
    norm opens, and the residual gradient is folded into the norm's backward kernel.  ``passthrough=2`` returns
    ``(y, x_res, x_skip)``: ``x_skip`` (== x again) is for a second consumer outside the block (a skip connection)."""
    return GroupNormFn.apply(x, gamma, beta, film, groups, eps, 1 if silu else 0, int(passthrough))


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _require_gpu(x)
        x = _c(x)
        D = x.shape[-1]
        R = x.numel() // D
        g32, b32 = _c(gamma.detach().float()), _c(beta.detach().float())
        y = torch.empty_like(x)
        stats = torch.empty((R, 2), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().mdm_ln_fwd(_p(x), _p(g32), _p(b32), _p(y), _p(stats), R, D, float(eps), _dt(x), _stream()), "mdm_ln_fwd")
        ctx.save_for_backward(x, gamma, beta, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        dy = _c(dy)
        D = x.shape[-1]
        R = x.numel() // D
        g32 = _c(gamma.detach().float())
        dx = torch.empty_like(x)
        sg, sb = _slot(gamma), _slot(beta)
        sunk = sg is not None and sb is not None
        dgamma = sg if sunk else torch.empty(D, dtype=torch.float32, device=x.device)
        dbeta = sb if sunk else torch.empty(D, dtype=torch.float32, device=x.device)
        ws = torch.empty(((R + 63) // 64) * D * 2, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().mdm_ln_bwd(_p(dy), _p(x), _p(g32), _p(stats), _p(dx), _p(dgamma), _p(dbeta), _p(ws), R, D,
                                         1 if sunk else 0, _dt(x), _stream()), "mdm_ln_bwd")
        if sunk:
            _grad_sink.ready(gamma)
            _grad_sink.ready(beta)
            return dx, None, None, None
        return dx, dgamma, dbeta, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x, gamma, beta, eps)


# --------------------------------------------------------------------------------------
# text key / value projections of ALL attention layers at once
# --------------------------------------------------------------------------------------
def _ptr_array(ts):
    return (ctypes.c_void_p * len(ts))(*[_p(t) for t in ts])


class TextKVFn(torch.autograd.Function):
    """kvc_l = Linear_l(LayerNorm_l(cond)) for every attention layer l (reference unet.py:263-264, 304: each layer
    normalises and projects the SAME text states).  Forward: one multi-LayerNorm launch + one grouped GEMM launch per
    output width; backward: grouped input-gradient GEMMs, one multi-LayerNorm backward (which also sums the L gradients
    of ``cond``), grouped weight gradients.  ~8 launches instead of ~8 per layer; bf16 only (the fp32 parity mode runs
    the layers one by one)."""

    @staticmethod
    def forward(ctx, cond, eps, *params):
        _require_gpu(cond)
        cond = _c(cond)
        B, S, D = cond.shape
        R, L = B * S, len(params) // 4
        lnw, lnb, ws, bs = params[0::4], params[1::4], params[2::4], params[3::4]
        lib = _lib.lib()
        g32 = [_c(t.detach().float()) for t in lnw]
        b32 = [_c(t.detach().float()) for t in lnb]
        cn = [torch.empty_like(cond) for _ in range(L)]
        stats = torch.empty((R, 2), dtype=torch.float32, device=cond.device)
        _lib.check(lib.mdm_ln_multi_fwd(_p(cond), _ptr_array(g32), _ptr_array(b32), _ptr_array(cn), L, _p(stats), R, D, float(eps),
                                        _dt(cond), _stream()), "mdm_ln_multi_fwd")
        groups = {}
        for l, w in enumerate(ws):
            groups.setdefault(w.shape[0], []).append(l)
        ys = [None] * L
        for cout, idx in groups.items():
            packs = [packed_weight(ws[l], bs[l], cond.dtype) for l in idx]
            for l in idx:
                ys[l] = torch.empty((B, S, cout), dtype=cond.dtype, device=cond.device)
            _prof_wrap("linear grouped x%d M=%d N=%d K=%d" % (len(idx), R, cout, D), 2.0 * len(idx) * R * cout * D, lambda: _lib.check(
                lib.mdm_linear_grouped(_ptr_array([cn[l] for l in idx]), _ptr_array([pk[0] for pk in packs]),
                                       _ptr_array([pk[2] for pk in packs]), _ptr_array([ys[l] for l in idx]), len(idx), R, D, cout,
                                       _dt(cond), _stream()), "mdm_linear_grouped"))
        ctx.save_for_backward(cond, stats, *cn, *params)
        ctx.L, ctx.groups = L, groups
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        saved = ctx.saved_tensors
        L, groups = ctx.L, ctx.groups
        cond, stats, cn, params = saved[0], saved[1], saved[2:2 + L], saved[2 + L:]
        lnw, lnb, ws, bs = params[0::4], params[1::4], params[2::4], params[3::4]
        B, S, D = cond.shape
        R = B * S
        lib = _lib.lib()
        dys = [_c(d) if d is not None else torch.zeros((B, S, ws[l].shape[0]), dtype=cond.dtype, device=cond.device)
               for l, d in enumerate(dys)]
        dcn = [torch.empty_like(cond) for _ in range(L)]
        grads = [None] * (4 * L)
        def dgrad(cout, idx):
            packs = [packed_weight(ws[l], bs[l], cond.dtype) for l in idx]
            _prof_wrap("linear grouped x%d M=%d N=%d K=%d" % (len(idx), R, D, cout), 2.0 * len(idx) * R * cout * D, lambda: _lib.check(
                lib.mdm_linear_grouped(_ptr_array([dys[l] for l in idx]), _ptr_array([pk[1] for pk in packs]), None,
                                       _ptr_array([dcn[l] for l in idx]), len(idx), R, cout, D, _dt(cond), _stream()),
                "mdm_linear_grouped"))

        for cout, idx in groups.items():
            if not _late_wgrad_first:   # (see SharedInputLinearsFn.backward: the weight gradients are handed over first)
                dgrad(cout, idx)
            # weight / bias gradients: one grouped launch, into the gradient arena when there is one
            slots = [(_slot(ws[l]), _slot(bs[l])) for l in idx]
            sunk = all(a is not None and b is not None for a, b in slots)
            if sunk:
                dws, dbs = [a for a, _ in slots], [b for _, b in slots]
            else:
                dws = [torch.zeros(ws[l].shape, dtype=torch.float32, device=cond.device) for l in idx]
                dbs = [torch.zeros(bs[l].shape, dtype=torch.float32, device=cond.device) for l in idx]
            xs2, dy2 = [cn[l].reshape(R, D) for l in idx], [dys[l].reshape(R, cout) for l in idx]

            def go(xs2=xs2, dy2=dy2, dws=dws, dbs=dbs, idx=idx, sunk=sunk, cout=cout):
                _prof_wrap("conv_wgrad grouped x%d M=%d N=%d K=%d" % (len(idx), R, cout, D), 2.0 * len(idx) * R * cout * D,
                           lambda: wgrad_grouped(xs2, dy2, dws, dbs))
                if sunk:
                    for l in idx:
                        _grad_sink.ready(ws[l])
                        _grad_sink.ready(bs[l])

            if sunk:
                _off_critical_path(xs2 + dy2, go)
            else:
                go()
                for k, l in enumerate(idx):
                    grads[4 * l + 2], grads[4 * l + 3] = dws[k], dbs[k]
        if _late_wgrad_first:
            for cout, idx in groups.items():
                dgrad(cout, idx)
        g32 = [_c(t.detach().float()) for t in lnw]
        nslots = [(_slot(lnw[l]), _slot(lnb[l])) for l in range(L)]
        nsunk = all(a is not None and b is not None for a, b in nslots)
        if nsunk:
            dgs, dbs_ = [a for a, _ in nslots], [b for _, b in nslots]
        else:
            dgs = [torch.empty(D, dtype=torch.float32, device=cond.device) for _ in range(L)]
            dbs_ = [torch.empty(D, dtype=torch.float32, device=cond.device) for _ in range(L)]
        dcond = torch.empty_like(cond)
        _lib.check(lib.mdm_ln_multi_bwd(_ptr_array(dcn), _p(cond), _ptr_array(g32), _p(stats), _p(dcond), _ptr_array(dgs),
                                        _ptr_array(dbs_), L, R, D, 1 if nsunk else 0, _dt(cond), _stream()), "mdm_ln_multi_bwd")
        if nsunk:
            for l in range(L):
                _grad_sink.ready(lnw[l])
                _grad_sink.ready(lnb[l])
        else:
            for l in range(L):
                grads[4 * l], grads[4 * l + 1] = dgs[l], dbs_[l]
        return (dcond, None) + tuple(grads)


class SharedInputLinearsFn(torch.autograd.Function):
    """y_l = x W_l^T + b_l for several linear layers that read the SAME input (every ResNet's ``time_layer`` applied to
    silu(temb), reference unet.py:206, 227): one grouped GEMM per output width for the outputs, one for the input
    gradients, one grouped weight-gradient launch -- instead of three tiny launches per layer.  bf16 only."""

    @staticmethod
    def forward(ctx, x, *params):
        _require_gpu(x)
        x = _c(x)
        R, D = x.shape
        ws, bs = params[0::2], params[1::2]
        L = len(ws)
        lib = _lib.lib()
        groups = {}
        for l, w in enumerate(ws):
            groups.setdefault(w.shape[0], []).append(l)
        ys = [None] * L
        for cout, idx in groups.items():
            packs = [packed_weight(ws[l], bs[l], x.dtype) for l in idx]
            for l in idx:
                ys[l] = torch.empty((R, cout), dtype=x.dtype, device=x.device)
            _lib.check(lib.mdm_linear_grouped(_ptr_array([x] * len(idx)), _ptr_array([pk[0] for pk in packs]),
                                              _ptr_array([pk[2] for pk in packs]), _ptr_array([ys[l] for l in idx]), len(idx), R, D,
                                              cout, _dt(x), _stream()), "mdm_linear_grouped")
        ctx.save_for_backward(x, *params)
        ctx.groups = groups
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        x, params = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        ws, bs = params[0::2], params[1::2]
        L, groups = len(ws), ctx.groups
        R, D = x.shape
        lib = _lib.lib()
        dys = [_c(d) if d is not None else torch.zeros((R, ws[l].shape[0]), dtype=x.dtype, device=x.device) for l, d in enumerate(dys)]
        dxs = [torch.empty_like(x) for _ in range(L)]
        grads = [None] * (2 * L)
        def dgrad(cout, idx):
            packs = [packed_weight(ws[l], bs[l], x.dtype) for l in idx]
            _lib.check(lib.mdm_linear_grouped(_ptr_array([dys[l] for l in idx]), _ptr_array([pk[1] for pk in packs]), None,
                                              _ptr_array([dxs[l] for l in idx]), len(idx), R, cout, D, _dt(x), _stream()),
                       "mdm_linear_grouped")

        # These layers' gradients arrive when backward ENDS: nothing of the main stream is left to hide behind.  The weight
        # gradients (side stream) are handed over BEFORE the input-gradient launches, so that they wait for what precedes
        # those, not for them (the hand-off makes the side stream wait for everything queued on the main stream so far).
        for cout, idx in groups.items():
            if not _late_wgrad_first:
                dgrad(cout, idx)
            slots = [(_slot(ws[l]), _slot(bs[l])) for l in idx]
            sunk = all(a is not None and b is not None for a, b in slots)
            if sunk:
                dws, dbs = [a for a, _ in slots], [b for _, b in slots]
            else:
                dws = [torch.zeros(ws[l].shape, dtype=torch.float32, device=x.device) for l in idx]
                dbs = [torch.zeros(bs[l].shape, dtype=torch.float32, device=x.device) for l in idx]
            xs2, dy2 = [x] * len(idx), [dys[l] for l in idx]

            def go(xs2=xs2, dy2=dy2, dws=dws, dbs=dbs, idx=idx, sunk=sunk):
                wgrad_grouped(xs2, dy2, dws, dbs)
                if sunk:
                    for l in idx:
                        _grad_sink.ready(ws[l])
                        _grad_sink.ready(bs[l])

            if sunk:
                _off_critical_path([x] + dy2, go)
            else:
                go()
                for k, l in enumerate(idx):
                    grads[2 * l], grads[2 * l + 1] = dws[k], dbs[k]
        if _late_wgrad_first:
            for cout, idx in groups.items():
                dgrad(cout, idx)
        dx = dxs[0] if L == 1 else torch.stack(dxs).sum(0)
        return (dx,) + tuple(grads)


def shared_input_linears_supported(x, layers):
    return x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[-1] % 64 == 0 and 0 < len(layers) and \
        all(w.shape[0] % 64 == 0 and b is not None for w, b in layers) and \
        max(sum(1 for w2, _ in layers if w2.shape[0] == w.shape[0]) for w, _ in layers) <= 32


def shared_input_linears(x, layers):
    """layers: [(weight [Cout_l, D], bias [Cout_l])] -> tuple of x W_l^T + b_l"""
    return SharedInputLinearsFn.apply(x, *[t for pair in layers for t in pair])


def text_kv_supported(cond, layers):
    """the grouped path needs bf16 text states of a width the buffer-addressed GEMM takes"""
    return cond.is_cuda and cond.dtype == torch.bfloat16 and cond.shape[-1] % 64 == 0 and 0 < len(layers) <= 32 and \
        all(w.shape[0] % 64 == 0 for _, _, w, _ in layers)


def text_kv(cond, layers, eps=1e-5):
    """layers: [(ln_weight, ln_bias, weight, bias)] -> tuple of kvc tensors [B, S, Cout_l]"""
    flat = [t for quad in layers for t in quad]
    return TextKVFn.apply(cond, eps, *flat)


# --------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------
class AttentionFn(torch.autograd.Function):
    """out = softmax(q k^T / sqrt(d)) v + softmax(q k_c^T / sqrt(d) [masked]) v_c."""

    @staticmethod
    def forward(ctx, qkv, kvc, mask, heads, keep):
        _require_gpu(qkv)
        qkv, kvc = _c(qkv), _c(kvc)
        B = qkv.shape[0]
        C = qkv.shape[-1] // 3
        L = qkv.numel() // (B * 3 * C)
        S = kvc.shape[1] if kvc is not None else 0
        d = C // heads
        m32 = _c(mask.float()) if mask is not None else None
        out = torch.empty(qkv.shape[:-1] + (C,), dtype=qkv.dtype, device=qkv.device)
        lse_s = torch.empty((B, heads, L), dtype=torch.float32, device=qkv.device) if keep else None
        lse_c = torch.empty((B, heads, L), dtype=torch.float32, device=qkv.device) if (keep and kvc is not None) else None
        oc = torch.empty_like(out) if kvc is not None else None   # also the kernel's staging buffer for the cross part
        _prof_wrap("attn_fwd_kernel<d=%d> L=%d" % (d, L), 4.0 * B * heads * L * (L + S) * d, lambda: _lib.check(
            _lib.lib().mdm_attn_fwd(_p(qkv), _p(kvc), _p(m32), _p(out), _p(oc), _p(lse_s), _p(lse_c), B, L, S, heads, d, _dt_mm(qkv), _stream()),
            "mdm_attn_fwd",
        ), kind="attn")
        ctx.save_for_backward(qkv, kvc, m32, out, oc, lse_s, lse_c)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, kvc, m32, out, oc, lse_s, lse_c = ctx.saved_tensors
        dout = _c(dout)
        B = qkv.shape[0]
        C = qkv.shape[-1] // 3
        L = qkv.numel() // (B * 3 * C)
        S = kvc.shape[1] if kvc is not None else 0
        heads = ctx.heads
        d = C // heads
        dqkv = torch.empty_like(qkv)
        dkvc = torch.empty_like(kvc) if kvc is not None else None
        delta_s = torch.empty_like(lse_s)
        delta_c = torch.empty_like(lse_c) if kvc is not None else None
        _prof_wrap("attn_bwd (delta + dq + dkv kernels)<d=%d> L=%d" % (d, L), 8.0 * B * heads * L * (L + S) * d, lambda: _lib.check(
            _lib.lib().mdm_attn_bwd(_p(qkv), _p(kvc), _p(m32), _p(out), _p(oc), _p(dout), _p(lse_s), _p(lse_c), _p(delta_s),
                                    _p(delta_c), _p(dqkv), _p(dkvc), B, L, S, heads, d, _dt(qkv), _stream()),
            "mdm_attn_bwd",
        ), kind="attn")
        return dqkv, dkvc, None, None, None


def attention(qkv, kvc, mask, heads):
    return AttentionFn.apply(qkv, kvc, mask, heads, torch.is_grad_enabled())


# --------------------------------------------------------------------------------------
# streaming helpers
# --------------------------------------------------------------------------------------
class ToNHWCFn(torch.autograd.Function):
    """fp32 NCHW (reference layout) -> compute-dtype NHWC with zero channel padding."""

    @staticmethod
    def forward(ctx, x, dtype, cpad):
        _require_gpu(x)
        x = _c(x.float())
        N, C, H, W = x.shape
        y = torch.empty((N, H, W, cpad), dtype=dtype, device=x.device)
        _lib.check(_lib.lib().mdm_nchw_to_nhwc(_p(x), _p(y), N, C, H, W, cpad, _dt(y), _stream()), "mdm_nchw_to_nhwc")
        ctx.C = C
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        N, H, W, Cs = dy.shape
        dx = torch.empty((N, ctx.C, H, W), dtype=torch.float32, device=dy.device)
        _lib.check(_lib.lib().mdm_nhwc_to_nchw(_p(dy), _p(dx), N, ctx.C, H, W, Cs, _dt(dy), _stream()), "mdm_nhwc_to_nchw")
        return dx, None, None


class FromNHWCFn(torch.autograd.Function):
    """compute-dtype NHWC (first C channels) -> fp32 NCHW."""

    @staticmethod
    def forward(ctx, x, C):
        _require_gpu(x)
        x = _c(x)
        N, H, W, Cs = x.shape
        y = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().mdm_nhwc_to_nchw(_p(x), _p(y), N, C, H, W, Cs, _dt(x), _stream()), "mdm_nhwc_to_nchw")
        ctx.Cs, ctx.dtype = Cs, x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy.float())
        N, C, H, W = dy.shape
        dx = torch.empty((N, H, W, ctx.Cs), dtype=ctx.dtype, device=dy.device)
        _lib.check(_lib.lib().mdm_nchw_to_nhwc(_p(dy), _p(dx), N, C, H, W, ctx.Cs, _dt(dx), _stream()), "mdm_nchw_to_nhwc")
        return dx, None


def to_nhwc(x_nchw, dtype, cpad=None):
    epv = 4 if dtype == torch.float32 else 8
    cpad = _round_up(x_nchw.shape[1], epv) if cpad is None else cpad
    return ToNHWCFn.apply(x_nchw, dtype, cpad)


def from_nhwc(x, C):
    return FromNHWCFn.apply(x, C)


class ConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a)
        a, b = _c(a), _c(b)
        C1, C2 = a.shape[-1], b.shape[-1]
        M = a.numel() // C1
        out = torch.empty(a.shape[:-1] + (C1 + C2,), dtype=a.dtype, device=a.device)
        _prof_wrap("concat", 2.0 * out.numel() * out.element_size(),
                   lambda: _lib.check(_lib.lib().mdm_concat(_p(a), _p(b), _p(out), M, C1, C2, 0, _dt(a), _stream()), "mdm_concat"), kind="hbm")
        ctx.C1, ctx.C2 = C1, C2
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _c(dout)
        C1, C2 = ctx.C1, ctx.C2
        M = dout.numel() // (C1 + C2)
        da = torch.empty(dout.shape[:-1] + (C1,), dtype=dout.dtype, device=dout.device)
        db = torch.empty(dout.shape[:-1] + (C2,), dtype=dout.dtype, device=dout.device)
        _lib.check(_lib.lib().mdm_concat(_p(da), _p(db), _p(dout), M, C1, C2, 1, _dt(dout), _stream()), "mdm_concat")
        return da, db


def concat(a, b):
    return ConcatFn.apply(a, b)


class Upsample2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        x = _c(x)
        N, H, W, C = x.shape
        y = torch.empty((N, 2 * H, 2 * W, C), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().mdm_upsample2x(_p(x), _p(y), N, H, W, C, _dt(x), _stream()), "mdm_upsample2x")
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        N, H2, W2, C = dy.shape
        dx = torch.empty((N, H2 // 2, W2 // 2, C), dtype=dy.dtype, device=dy.device)
        _lib.check(_lib.lib().mdm_downsum2x(_p(dy), _p(dx), N, H2 // 2, W2 // 2, C, _dt(dy), _stream()), "mdm_downsum2x")
        return dx


def upsample2x(x):
    return Upsample2xFn.apply(x)


def _ew(a, b, op):
    out = torch.empty_like(a)
    _lib.check(_lib.lib().mdm_elementwise(_p(a), _p(b), _p(out), a.numel(), op, _dt(a), _stream()), "mdm_elementwise")
    return out


class SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        x = _c(x)
        ctx.save_for_backward(x)
        return _ew(x, None, 0)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _ew(x, _c(dy), 1)


def silu(x):
    return SiluFn.apply(x)


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a)
        return _ew(_c(a), _c(b), 2)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    if a.shape != b.shape or a.dtype != b.dtype:
        raise _lib.MdmHipError("add: shape/dtype mismatch %s %s vs %s %s" % (a.shape, a.dtype, b.shape, b.dtype))
    return AddFn.apply(a, b)


class CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        _require_gpu(x)
        x = _c(x)
        ctx.src = x.dtype
        if x.dtype == dtype:
            return x.view_as(x)
        y = torch.empty(x.shape, dtype=dtype, device=x.device)
        _lib.check(_lib.lib().mdm_cast(_p(x), _p(y), x.numel(), _dt(x), _dt(y), _stream()), "mdm_cast")
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy.dtype == ctx.src:
            return dy, None
        dy = _c(dy)
        dx = torch.empty(dy.shape, dtype=ctx.src, device=dy.device)
        _lib.check(_lib.lib().mdm_cast(_p(dy), _p(dx), dy.numel(), _dt(dy), _dt(dx), _stream()), "mdm_cast")
        return dx, None


def cast(x, dtype):
    return CastFn.apply(x, dtype)


def sincos_embedding(times_f32, freqs_f32, dtype):
    """[B] x [half] -> [B, 2*half] = (sin | cos); no gradient (unet.py:835-836)."""
    _require_gpu(times_f32)
    t = _c(times_f32.detach().float())
    f = _c(freqs_f32.detach().float().reshape(-1))
    B, half = t.numel(), f.numel()
    out = torch.empty((B, 2 * half), dtype=dtype, device=t.device)
    _lib.check(_lib.lib().mdm_sincos_emb(_p(t), _p(f), _p(out), B, half, _dt(out), _stream()), "mdm_sincos_emb")
    return out


class MaskedMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        _require_gpu(x)
        x = _c(x)
        B, S, D = x.shape
        m32 = _c(mask.float()) if mask is not None else None
        y = torch.empty((B, D), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().mdm_masked_mean(_p(x), _p(m32), _p(y), B, S, D, _dt(x), _stream()), "mdm_masked_mean")
        ctx.save_for_backward(m32)
        ctx.shape = (B, S, D)
        return y

    @staticmethod
    def backward(ctx, dy):
        (m32,) = ctx.saved_tensors
        dy = _c(dy)
        B, S, D = ctx.shape
        dx = torch.empty((B, S, D), dtype=dy.dtype, device=dy.device)
        _lib.check(_lib.lib().mdm_masked_mean_bwd(_p(dy), _p(m32), _p(dx), B, S, D, 0, _dt(dy), _stream()), "mdm_masked_mean_bwd")
        return dx, None


def masked_mean(x, mask):
    return MaskedMeanFn.apply(x, mask)


# --------------------------------------------------------------------------------------
# per-pixel arithmetic around the denoiser: NCHW fp32 images (row a7 x / std)
# --------------------------------------------------------------------------------------
def _img(t):
    """fp32, contiguous, on the GPU"""
    _require_gpu(t)
    t = t.detach()
    if t.dtype != torch.float32:
        raise _lib.MdmHipError("image-side ops take fp32 tensors (got %s)" % t.dtype)
    return _c(t)


def _vec(t, B):
    """per-sample scalars ([B], [B,1,1,1] ...) -> contiguous fp32 [B] on the GPU"""
    t = _c(t.detach().reshape(-1).float())
    if t.numel() != B:
        raise _lib.MdmHipError("expected %d per-sample values, got %d" % (B, t.numel()))
    return t


class DeviceRng:
    """{seed, offset} of the library's counter-based normal generator (Philox4x32-10 + Box-Muller), kept ON THE DEVICE so
    that draws and advances are plain stream work (capturable in a hipGraph).  Element i of a draw is lane i & 3 of
    counter block offset + i // 4; oracle/philox_ref.py regenerates the same numbers on the host."""

    def __init__(self, seed: int, device, offset: int = 0):
        self.state = torch.tensor([seed & (2**63 - 1), offset], dtype=torch.int64, device=device)

    def advance(self, n_elements: int):
        _lib.check(_lib.lib().mdm_rng_advance(_p(self.state), (int(n_elements) + 3) // 4, _stream()), "mdm_rng_advance")

    def randn(self, shape, stream_id: int = 0):
        out = torch.empty(shape, dtype=torch.float32, device=self.state.device)
        n = out.numel()
        if n % 4:
            raise _lib.MdmHipError("randn: element count must be a multiple of 4")
        _lib.check(_lib.lib().mdm_randn(_p(out), n, _p(self.state), stream_id, _stream()), "mdm_randn")
        self.advance(n)
        return out


_PT = {"DDPM": 0, "DDIM": 1, "V_PREDICTION": 2}


def _ptype(pt):
    return _PT[pt.name] if hasattr(pt, "name") else int(pt)


def sampler_step(x_t, pred, g, g_last, prediction_type, ddim_eta=None, need_noise=False, noise=None, rng=None,
                 rng_stream=0, clip="CLIP", image_scale=1.0, pred_uncond=None, guidance_scale=1.0, thr=None,
                 noise_gate=None):
    """One reverse-diffusion update (reference samplers.py:281-345 + 445-456 + 500-508) as ONE kernel.
    -> (x0, x_last).  ``clip``: "NONE" | "CLIP" | "DYNAMIC" (needs ``thr`` [B]) | "X0_ONLY" (returns the unclipped
    x0 * image_scale and None: the input of the dynamic-threshold quantile)."""
    x_t, pred = _img(x_t), _img(pred)
    B = x_t.shape[0]
    chw = x_t.numel() // B
    g, gl = _vec(g, B), _vec(g_last, B)
    pu = _img(pred_uncond) if pred_uncond is not None else None
    nz = _img(noise) if noise is not None else None
    th = _vec(thr, B) if thr is not None else None
    gate = _c(noise_gate.detach().reshape(-1).float()) if noise_gate is not None else None
    cm = {"NONE": 0, "CLIP": 1, "DYNAMIC": 2, "X0_ONLY": 3}[clip]
    x0 = torch.empty_like(x_t)
    xl = torch.empty_like(x_t) if cm != 3 else None
    mode, eta = (0, 0.0) if ddim_eta is None else (1, float(ddim_eta))
    gen = bool(need_noise) and nz is None and not (mode == 1 and eta <= 0)
    if gen and rng is None:
        raise _lib.MdmHipError("sampler_step: need_noise without noise= or rng=")
    _lib.check(
        _lib.lib().mdm_sampler_step(_p(x_t), _p(pred), _p(pu), float(guidance_scale), _p(g), _p(gl), _p(nz), _p(gate), _p(th),
                                    _p(rng.state) if rng is not None else None, int(rng_stream), _p(x0), _p(xl), B, chw,
                                    _ptype(prediction_type), mode, eta, 1 if need_noise else 0, cm,
                                    float(image_scale) if image_scale else 1.0, _stream()),
        "mdm_sampler_step",
    )
    return x0, xl


def noise_images(images, g, eps=None, rng=None, rng_stream=0, inv_scale=1.0):
    """x_t = sqrt(g) * images * inv_scale + sqrt(1 - g) * eps (reference samplers.py:244-246).  ``eps=None`` draws the
    noise inside the kernel from ``rng``.  -> (x_t, eps)"""
    images = _img(images)
    B = images.shape[0]
    chw = images.numel() // B
    g = _vec(g, B)
    x_t = torch.empty_like(images)
    if eps is None:
        if rng is None:
            raise _lib.MdmHipError("noise_images: give eps= or rng=")
        eps_out = torch.empty_like(images)
        _lib.check(_lib.lib().mdm_noise_images(_p(images), None, _p(g), float(inv_scale), _p(x_t), _p(eps_out), _p(rng.state),
                                               int(rng_stream), B, chw, _stream()), "mdm_noise_images")
        return x_t, eps_out
    eps = _img(eps)
    _lib.check(_lib.lib().mdm_noise_images(_p(images), _p(eps), _p(g), float(inv_scale), _p(x_t), None, None, 0, B, chw,
                                           _stream()), "mdm_noise_images")
    return x_t, eps


class DiffusionLossFn(torch.autograd.Function):
    """loss[b] = mean_chw (to_target_space(pred; x_t, g) - target(images, eps, g))^2 (reference diffusion.py:144-168
    with samplers.py:266-279, 347-390 folded in); differentiable w.r.t. ``pred`` only."""

    @staticmethod
    def forward(ctx, pred, x_t, images, eps, g, inv_scale, ptype, ttype):
        pred, x_t, images, eps = _img(pred), _img(x_t), _img(images), _img(eps)
        B = pred.shape[0]
        chw = pred.numel() // B
        g = _vec(g, B)
        wsb = ctypes.c_size_t(0)
        _lib.check(_lib.lib().mdm_diffusion_loss_plan(B, chw, ctypes.byref(wsb)), "mdm_diffusion_loss_plan")
        ws = _f32_ws(wsb.value, pred.device)
        loss = torch.empty(B, dtype=torch.float32, device=pred.device)
        _lib.check(_lib.lib().mdm_diffusion_loss_fwd(_p(x_t), _p(pred), _p(images), _p(eps), _p(g), float(inv_scale), _p(loss),
                                                     _p(ws), B, chw, ptype, ttype, _stream()), "mdm_diffusion_loss_fwd")
        ctx.save_for_backward(pred, x_t, images, eps, g)
        ctx.cfg = (float(inv_scale), ptype, ttype)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        pred, x_t, images, eps, g = ctx.saved_tensors
        inv_scale, ptype, ttype = ctx.cfg
        B = pred.shape[0]
        chw = pred.numel() // B
        gl = _vec(gloss, B)
        dpred = torch.empty_like(pred)
        _lib.check(_lib.lib().mdm_diffusion_loss_bwd(_p(x_t), _p(pred), _p(images), _p(eps), _p(g), _p(gl), inv_scale, _p(dpred),
                                                     B, chw, ptype, ttype, _stream()), "mdm_diffusion_loss_bwd")
        return dpred, None, None, None, None, None, None, None


def diffusion_loss(pred, x_t, images, eps, g, prediction_type, target_type, inv_scale=1.0):
    if pred.dtype != torch.float32:
        raise _lib.MdmHipError("diffusion_loss: the model output at the boundary is fp32 (got %s)" % pred.dtype)
    return DiffusionLossFn.apply(pred, x_t, images, eps, g, float(inv_scale), _ptype(prediction_type), _ptype(target_type))


def avgpool(x, r):
    """F.avg_pool2d(x, r) on an NCHW fp32 image batch (the training pyramid, reference diffusion.py:346-348); no gradient"""
    x = _img(x)
    N, C, H, W = x.shape
    y = torch.empty((N, C, H // r, W // r), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mdm_avgpool(_p(x), _p(y), N, C, H, W, int(r), _stream()), "mdm_avgpool")
    return y


class SampleStdFn(torch.autograd.Function):
    """y = x / x.std((1, 2, 3), keepdim=True) (unbiased): the input normalisation of a nested level
    (reference models/unet.py:871-872)."""

    @staticmethod
    def forward(ctx, x):
        x = _img(x)
        N = x.shape[0]
        chw = x.numel() // N
        y = torch.empty_like(x)
        stats = torch.empty((N, 2), dtype=torch.float32, device=x.device)
        ws = torch.empty(N * 64 * 2, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().mdm_sample_std_fwd(_p(x), _p(y), _p(stats), _p(ws), N, chw, _stream()), "mdm_sample_std_fwd")
        ctx.save_for_backward(x, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats = ctx.saved_tensors
        dy = _c(dy.float())
        N = x.shape[0]
        chw = x.numel() // N
        dx = torch.empty_like(x)
        ws = torch.empty(N * 64 * 2, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().mdm_sample_std_bwd(_p(dy), _p(x), _p(stats), _p(dx), _p(ws), N, chw, _stream()), "mdm_sample_std_bwd")
        return dx


def sample_std_normalize(x):
    return SampleStdFn.apply(x)


def input_stage(u8_nhwc):
    """uint8 NHWC [B, H, W, 3] -> fp32 NCHW, (u - 127) / 128 (reference clis/train_parallel.py:194-195)"""
    _require_gpu(u8_nhwc)
    if u8_nhwc.dtype != torch.uint8 or u8_nhwc.dim() != 4 or u8_nhwc.shape[-1] != 3:
        raise _lib.MdmHipError("input_stage: expected a uint8 [B, H, W, 3] tensor")
    u = _c(u8_nhwc)
    B, H, W, _ = u.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=u.device)
    _lib.check(_lib.lib().mdm_input_stage(_p(u), _p(out), B, H, W, _stream()), "mdm_input_stage")
    return out


# --------------------------------------------------------------------------------------
# optimizer tail
# --------------------------------------------------------------------------------------
def sumsq(flat_f32, out=None, step_counter=None):
    """device scalar sum(g^2) of a flat fp32 arena (no host sync); ``step_counter`` (device int32[1]) += 1 when finite"""
    _require_gpu(flat_f32)
    out = torch.empty(1, dtype=torch.float32, device=flat_f32.device) if out is None else out
    ws = torch.empty(1024, dtype=torch.float32, device=flat_f32.device)
    _lib.check(_lib.lib().mdm_sumsq(_p(flat_f32), _p(out), _p(ws), flat_f32.numel(), _p(step_counter), _stream()), "mdm_sumsq")
    return out


def adamw_ema_step(p, g, m, v, ema, gnorm_sq, lr, beta1, beta2, eps, weight_decay, step, clip, ema_decay, zero_grad=True,
                   step_dev=None):
    """one fused clip + AdamW + EMA (+ zero-grad) pass over flat fp32 arenas; invalidates the packed-weight cache.
    ``step_dev`` (device int32[1]) overrides ``step`` for the bias correction."""
    _require_gpu(p)
    nbytes = 4.0 * p.numel() * ((4 if ema is None else 5) + (3 if ema is None else 4) + (1 if zero_grad else 0))
    _prof_wrap("adamw_ema_step", nbytes, lambda: _lib.check(
        _lib.lib().mdm_adamw_ema_step(_p(p), _p(g), _p(m), _p(v), _p(ema), _p(gnorm_sq), p.numel(), float(lr), float(beta1),
                                      float(beta2), float(eps), float(weight_decay), int(step), _p(step_dev), float(clip), float(ema_decay),
                                      1 if zero_grad else 0, _stream()),
        "mdm_adamw_ema_step",
    ), kind="hbm")
    invalidate_packed_weights()


# --------------------------------------------------------------------------------------
# dropout (nn.Dropout of a ResNet block, reference models/unet.py:208,234)
# --------------------------------------------------------------------------------------
_dropout_rng = [None, 0]   # [seed, next Philox counter block]; seeded from torch's generator on first use


def seed_dropout(seed: int, rank: int = None):
    """restart the dropout mask stream.  Default seed (first use): ``torch.initial_seed()``; the data-parallel rank is
    folded in (``rank`` default: torch.distributed's, else 0), so ranks that share a seed -- the reference seeds every
    rank alike, clis/train_parallel.py -- still draw different masks for their different samples."""
    if rank is None:
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
    _dropout_rng[0] = (int(seed) + 0x9E3779B97F4A7C15 * int(rank)) & 0xFFFFFFFFFFFFFFFF
    _dropout_rng[1] = 0


def get_dropout_rng_state():
    """(seed, offset) of the dropout mask stream -- not part of torch's RNG state: save it next to
    ``torch.cuda.get_rng_state()`` in a checkpoint if bit-reproducible resumption with dropout > 0 matters"""
    if _dropout_rng[0] is None:
        seed_dropout(torch.initial_seed())
    return (_dropout_rng[0], _dropout_rng[1])


def set_dropout_rng_state(state):
    _dropout_rng[0], _dropout_rng[1] = int(state[0]) & 0xFFFFFFFFFFFFFFFF, int(state[1])


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        _require_gpu(x)
        x = _c(x)
        if _dropout_rng[0] is None:
            seed_dropout(torch.initial_seed())
        seed, off = _dropout_rng
        _dropout_rng[1] = off + (x.numel() + 3) // 4
        y = torch.empty_like(x)
        _lib.check(_lib.lib().mdm_dropout(_p(x), _p(y), x.numel(), float(p), seed, off, _dt(x), _stream()), "mdm_dropout")
        ctx.p, ctx.seed, ctx.off = float(p), seed, off
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        _lib.check(_lib.lib().mdm_dropout(_p(dy), _p(dx), dy.numel(), ctx.p, ctx.seed, ctx.off, _dt(dy), _stream()), "mdm_dropout")
        return dx, None


def dropout(x, p: float, training: bool = True):
    """``F.dropout(x, p, training)`` with a counter-based mask (Philox, regenerated in backward instead of stored).  The
    mask differs from torch's for the same seed -- as between any two generators -- the distribution is the same."""
    if not training or p <= 0.0:
        return x
    if p >= 1.0:
        return x * 0.0   # F.dropout(p = 1): everything dropped (the kernel's 1 / (1 - p) scale has no value there)
    if x.numel() % 8 != 0:
        raise _lib.MdmHipError("dropout: the element count must be a multiple of 8")
    return DropoutFn.apply(x, p)

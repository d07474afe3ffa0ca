"""The train step of the headline metric behind the reference's own entry point.

``train_batch(model, sample, optimizer, scheduler, logger, args, grad_scaler, accumulate_gradient,
num_grad_accumulations, ema_model, loss_factor)`` has the signature, the order of operations, the logging and the
return value of ``ml_mdm.trainer.train_batch`` (reference trainer.py:13-96), so ``clis/train_parallel.py`` runs
on it with one import swapped (``from mdm_hip import trainer``).  What it does underneath depends on the model:

* **fused path** -- the vision model is this package's UNet / NestedUNet on an MI355X, the optimizer is the
  ``torch.optim.AdamW`` / ``Adam`` the CLI builds (train_parallel.py:122-134) and the wrapper around ``model.model``
  is either none (one GPU) or ``mdm_hip.distributed.DataParallel``.  On the first call the step ADOPTS what the CLI
  built: the parameters move into one flat fp32 arena (same ``nn.Parameter`` objects, new storage), the optimizer's
  moments and the ``ModelEma`` copy's parameters become views of flat arenas as well -- so ``optimizer.state_dict()``,
  ``ema_model.save()`` and ``vision_model.save()`` keep working -- and from then on backward kernels ADD parameter
  gradients straight into a flat gradient arena (``ops.set_grad_sink``), and clip + AdamW + EMA + zero-grad are two
  streaming passes (``mdm_sumsq`` + ``mdm_adamw_ema_step``).  Gradient accumulation
  micro-steps (``accumulate_gradient=True`` under ``model.model.no_sync()``, train_parallel.py:196-230) simply keep
  adding into the arena; the learning rate is read from ``optimizer.param_groups`` every step, so any scheduler works.
* **plain path** -- anything else (CPU tensors, torch's DistributedDataParallel, another optimizer): the reference's
  sequence with torch's own pieces -- autograd accumulation, ``clip_grad_norm_``, ``optimizer.step()``,
  ``ema_model.update()`` -- the denoiser's kernels are still this package's.  ``bench.py --reference-loop`` times it.

bf16 autocast needs no loss scaling: the reference's ``GradScaler`` (trainer.py:43-53) multiplies the loss by a
power of two and divides the gradients by it again, which is exact in bf16/fp32 short of overflow; the fused path
therefore leaves the scaler alone (the plain path drives it like the reference does).

``TrainStep`` is the same fused step as a self-contained object (tests, tools).
"""
import collections
import math
import os

import torch

from . import ops
from .distributed import DataParallel, GradReducer


class FusedState:
    """Flat fp32 arenas for parameters / gradients / Adam moments / EMA of one vision model, and the fused optimizer
    tail over them.  Arena order = reverse registration order (the order backward produces gradients)."""

    def __init__(self, net, reducer=None, bucket_mb=256.0, wire_dtype="auto", use_ema=True, async_wgrad=True):
        self.net = net
        self.params = [p for p in net.parameters() if p.requires_grad]
        late = net.late_gradient_parameters() if hasattr(net, "late_gradient_parameters") else None
        self.reducer = reducer if reducer is not None else GradReducer(self.params, bucket_mb=bucket_mb, wire_dtype=wire_dtype,
                                                                       late_params=late)
        self.reducer.broadcast_parameters(0)
        flat = torch.empty_like(self.reducer.flat)
        self.slices = {}
        off = 0
        with torch.no_grad():
            for p in self.reducer.order:    # the arenas share the gradient arena's layout: the optimizer walks them in step
                n = p.numel()
                flat[off:off + n].copy_(p.reshape(-1))
                p.data = flat[off:off + n].view_as(p)
                self.slices[id(p)] = (off, n)
                off += n
        self.flat_p = flat
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.flat_ema = flat.clone() if use_ema else None
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=flat.device)   # optimizer step number, on the device
        self.reducer.rebind()
        self.async_wgrad = async_wgrad
        self._host_vals = {}     # name -> (pinned host scalar, event): device scalars the host reads AFTER it has queued more work
        self.activate()

    def activate(self):
        """route the backward kernels' parameter gradients into this state's arena"""
        ops.set_grad_sink(self.reducer)
        ops.enable_async_wgrad(self.async_wgrad)
        ops.enable_deferred_wgrad(self.async_wgrad, max_group=32 if self.reducer.world == 1 else 13)
        ops.invalidate_packed_weights()

    def view(self, arena, p):
        off, n = self.slices[id(p)]
        return arena[off:off + n].view_as(p)

    def finish_backward(self):
        """everything that has to be in the gradient arena before the optimizer reads it"""
        ops.flush_wgrad_queue()
        self.reducer.mark("bw1")
        ops.join_side_stream()
        self.reducer.finish()

    def post_scalar(self, name, value):
        """queue an asynchronous device -> pinned-host copy of a one-element tensor behind what the current stream holds NOW
        (and an event); `read_scalar(name)` later waits for exactly that point of the stream -- not for the kernels the host
        has queued since.  This is what lets train_batch queue backward + optimizer tail before it looks at the loss."""
        ent = self._host_vals.get(name)
        if ent is None:
            ent = (torch.empty(1, dtype=torch.float32).pin_memory(), torch.cuda.Event())
            self._host_vals[name] = ent
        ent[0].copy_(value.detach().reshape(1), non_blocking=True)
        ent[1].record()

    def read_scalar(self, name):
        buf, ev = self._host_vals[name]
        ev.synchronize()
        return float(buf[0])

    def post_skip_flag(self):
        """-> (pinned host scalar, event) that will hold THIS step's squared norm of the reduced gradient (non-finite = the
        device skipped the update).  A ring of buffers, because the flag of step N is looked at two calls later (`_settle`)
        while steps N + 1 and N + 2 post theirs."""
        ring = self.__dict__.setdefault("_skip_ring", [])
        k = self.__dict__.get("_skip_next", 0)
        if len(ring) <= k:
            ring.append((torch.empty(1, dtype=torch.float32).pin_memory(), torch.cuda.Event()))
        buf, ev = ring[k]
        self._skip_next = (k + 1) % 4
        buf.copy_(self.gnorm_sq.detach().reshape(1), non_blocking=True)
        ev.record()
        return buf, ev

    def optimizer_step(self, lr, betas, eps, weight_decay, clip_norm, ema_decay, dtype, loss=None):
        """clip + AdamW + EMA + zero-grad over the arenas (two launches) and the re-pack of the kernel-layout weights.
        A non-finite gradient norm skips the update, clears the gradients and does not advance the step number.
        `loss` (a device scalar, one rank only): a NaN loss makes the norm NaN whatever the gradients turned out to be --
        the reference's "NaN loss => no step" (trainer.py:38-41, 64-69) decided on the device, without the host."""
        if loss is not None and self.reducer.world == 1:
            # (into the arena, ahead of the norm: mdm_sumsq is also what advances the device-side step number)
            self.reducer.flat[:1].add_(loss.detach().float().reshape(1) * 0.0)   # NaN * 0 = NaN, finite * 0 = 0
        ops.sumsq(self.reducer.flat, out=self.gnorm_sq, step_counter=self.step_dev)
        ops.adamw_ema_step(self.flat_p, self.reducer.flat, self.m, self.v, self.flat_ema, self.gnorm_sq, lr, betas[0], betas[1],
                           eps, weight_decay, 0, clip_norm, ema_decay, zero_grad=True, step_dev=self.step_dev)
        ops.repack_all(dtype)   # one launch instead of one per weight


# --------------------------------------------------------------------------------------------------------------
# the reference's entry point
# --------------------------------------------------------------------------------------------------------------
def _vision_model(model):
    core = getattr(model.model, "module", model.model)
    return core.vision_model


def _adopt(model, optimizer, ema_model):
    """-> FusedState for (model, optimizer, ema_model), built on first use and cached on the optimizer; None when the
    combination does not qualify for the fused path (see the module docstring)."""
    st = getattr(optimizer, "_mdm_fused", None)
    if st is not None:
        if st is not False and st.ema_id != (id(ema_model) if ema_model is not None else None):
            # the EMA arena was wired (or left out) at adoption: a different / late ModelEma would silently never update
            raise RuntimeError("train_batch was first called %s and now %s: pass the same ema_model on every call "
                               "(the fused step adopted it together with the optimizer)"
                               % ("without an ema_model" if st.ema_id is None else "with another ema_model",
                                  "without one" if ema_model is None else "with one"))
        return st if st is not False else None

    def no(reason):
        optimizer._mdm_fused = False
        optimizer._mdm_fused_reason = reason
        return None

    from .unet import UNet

    wrapped = model.model
    net = _vision_model(model)
    if isinstance(wrapped, torch.nn.parallel.DistributedDataParallel):
        return no("torch DistributedDataParallel owns the gradient buckets (use mdm_hip.distributed.DataParallel)")
    if not isinstance(net, UNet):
        return no("vision model is not an mdm_hip UNet")
    params = [p for p in net.parameters() if p.requires_grad]
    if not params or not all(p.is_cuda and p.dtype == torch.float32 for p in params):
        return no("parameters are not fp32 tensors on the GPU")
    if type(optimizer) not in (torch.optim.AdamW, torch.optim.Adam) or len(optimizer.param_groups) != 1:
        return no("optimizer is not a single-group torch Adam / AdamW")
    grp = optimizer.param_groups[0]
    if grp.get("amsgrad") or grp.get("maximize") or (type(optimizer) is torch.optim.Adam and grp.get("weight_decay", 0) != 0
                                                      and not grp.get("decoupled_weight_decay", False)):
        return no("optimizer options without a fused counterpart (amsgrad / maximize / L2 weight decay)")
    if [id(p) for p in grp["params"]] != [id(p) for p in params]:
        return no("optimizer does not hold exactly the vision model's parameters")
    ema_net = None
    if ema_model is not None:
        ema_net = ema_model.module
        # ModelEma.update walks the state_dict: parameters AND persistent buffers (this package's nets have none)
        if getattr(ema_model, "device", None) is not None or set(net.state_dict()) != {n for n, _ in net.named_parameters()} or \
                [(n, tuple(p.shape)) for n, p in ema_net.named_parameters()] != [(n, tuple(p.shape)) for n, p in net.named_parameters()]:
            return no("ModelEma on another device / of another shape / with persistent buffers")
    reducer = wrapped.reducer if isinstance(wrapped, DataParallel) else \
        GradReducer(params, group=None, world_override=1)   # an unwrapped model trains locally, whatever torch.distributed holds
    if [id(p) for p in reducer.params] != [id(p) for p in params]:
        return no("the DataParallel wrapper's reducer does not hold exactly the vision model's parameters")
    had_state = {id(p): dict(optimizer.state.get(p, {})) for p in params}
    # MDM_HIP_SERIAL_WGRAD=1 (profiling aid, bench.py --serial-wgrad): weight gradients on the main stream
    st = FusedState(net, reducer=reducer, use_ema=ema_net is not None, async_wgrad=os.environ.get("MDM_HIP_SERIAL_WGRAD", "0") != "1")
    with torch.no_grad():
        # the optimizer's moments (a resumed run has them) and step count move into the arenas; the entries the
        # optimizer keeps are views, so optimizer.state_dict() / load_state_dict() round-trip as before
        steps = 0
        for p in params:
            old = had_state[id(p)]
            mv, vv = st.view(st.m, p), st.view(st.v, p)
            if "exp_avg" in old:
                mv.copy_(old["exp_avg"])
                vv.copy_(old["exp_avg_sq"])
                steps = max(steps, int(old["step"]))
            optimizer.state[p] = {"step": None, "exp_avg": mv, "exp_avg_sq": vv}
        st.step_dev.fill_(steps)
        # the step number lives on the device (a skipped non-finite step does not advance it); the optimizer's "step"
        # entries share one host tensor that is refreshed whenever somebody asks for the state_dict (checkpoint time)
        step_host = torch.tensor(float(steps))
        for p in params:
            optimizer.state[p]["step"] = step_host
        optimizer.register_state_dict_pre_hook(lambda opt, st=st, t=step_host: (_settle(st, keep=0), t.fill_(float(int(st.step_dev)))) and None)

        def _reinstall(opt, st=st, params=params, t=step_host):
            # optimizer.load_state_dict() replaced the state entries with fresh tensors: copy them into the arenas the
            # fused kernel reads and put the views (and the shared step tensor) back
            with torch.no_grad():
                steps_ = 0
                for p_ in params:
                    ent = opt.state.get(p_)
                    if not ent or "exp_avg" not in ent:
                        continue
                    mv_, vv_ = st.view(st.m, p_), st.view(st.v, p_)
                    if ent["exp_avg"].data_ptr() != mv_.data_ptr():
                        mv_.copy_(ent["exp_avg"])
                        vv_.copy_(ent["exp_avg_sq"])
                    steps_ = max(steps_, int(ent["step"]))
                    opt.state[p_] = {"step": t, "exp_avg": mv_, "exp_avg_sq": vv_}
                st.step_dev.fill_(steps_)
                t.fill_(float(steps_))

        optimizer.register_load_state_dict_post_hook(_reinstall)
        if ema_net is not None:
            by_name = dict(net.named_parameters())
            for name, pe in ema_net.named_parameters():
                ev = st.view(st.flat_ema, by_name[name])
                ev.copy_(pe.detach())
                pe.data = ev
    st.ema_id = id(ema_model) if ema_model is not None else None
    optimizer._mdm_fused = st
    return st


def _loss_of(losses, weights):
    return losses.mean() if weights is None else (losses * weights).sum() / weights.sum()


def train_batch(model, sample, optimizer, scheduler, logger, args, grad_scaler=None, accumulate_gradient=False,
                num_grad_accumulations=1, ema_model=None, loss_factor=1.0):
    """One micro-step of ``ml_mdm.trainer.train_batch`` (reference trainer.py:13-96): same arguments, same return value
    ``(loss_val, losses, times, x_t, means, targets)``; the caller wraps accumulation micro-steps in
    ``model.model.no_sync()`` exactly as train_parallel.py:196-214 does."""
    model.train()
    lr = scheduler.get_last_lr()[0]
    st = _adopt(model, optimizer, ema_model)
    if st is None and isinstance(model.model, DataParallel):
        # the plain path drives torch's own pieces (clip_grad_norm_, optimizer.step(), optimizer.zero_grad()): nothing in
        # it joins this wrapper's asynchronous all-reduces or keeps p.grad inside its arena -- ranks would diverge silently
        raise RuntimeError("mdm_hip.distributed.DataParallel only works with the fused train step, which this call cannot "
                           "take: %s.  Use torch.optim.AdamW / Adam (one parameter group, no amsgrad) with a ModelEma on "
                           "the model's device, or wrap the model in torch.nn.parallel.DistributedDataParallel instead."
                           % getattr(optimizer, "_mdm_fused_reason", "?"))
    fp16 = bool(getattr(args, "fp16", False))
    dev_type = "cuda" if next(_vision_model(model).parameters()).is_cuda else "cpu"
    if st is not None:
        return _train_batch_fused(st, model, sample, optimizer, scheduler, logger, args, accumulate_gradient,
                                  num_grad_accumulations, ema_model, loss_factor, lr, fp16)
    # ---- plain path: the reference's sequence with torch's own pieces ------------------------------------------------------
    if fp16:
        with torch.autocast(dev_type, dtype=torch.bfloat16):
            losses, times, x_t, means, targets, weights = model.get_loss(sample)
            loss = _loss_of(losses, weights) * loss_factor
            loss_val = loss.item()
            if math.isnan(loss_val):
                optimizer.zero_grad()
                return loss_val, losses, times, x_t, means, targets
            if num_grad_accumulations != 1:
                loss = loss / num_grad_accumulations
        if grad_scaler is not None:
            grad_scaler.scale(loss).backward()
        else:
            loss.backward()
    else:
        losses, times, x_t, means, targets, weights = model.get_loss(sample)
        loss = _loss_of(losses, weights)
        loss_val = loss.item()
        if math.isnan(loss_val):
            optimizer.zero_grad()
            optimizer.step()
            scheduler.step()
            return loss_val, losses, times, x_t, means, targets
        loss.backward()   # (the reference divides by num_grad_accumulations only AFTER this, trainer.py:73-75: no effect)

    if not accumulate_gradient:
        clip = float(getattr(args, "gradient_clip_norm", 2.0))
        core = getattr(model.model, "module", model.model)
        if fp16 and grad_scaler is not None:
            grad_scaler.unscale_(optimizer)
            torch.nn.utils.clip_grad_norm_(model.model.parameters(), clip)
            grad_scaler.step(optimizer)
            grad_scaler.update()
        else:
            torch.nn.utils.clip_grad_norm_(model.model.parameters(), clip)
            optimizer.step()
        ops.invalidate_packed_weights()
        if ema_model is not None:
            ema_model.update(core.vision_model)
        if logger is not None:
            logger.add_scalar("train/Loss", loss_val)
            logger.add_scalar("lr", lr)
        optimizer.zero_grad()
        scheduler.step()
    return loss_val, losses, times, x_t, means, targets


# the loss is read AFTER backward and the optimizer tail are queued (0: where the reference reads it, before backward --
# the host then cannot queue the backward until the forward has drained; A/B switch, HISTORY.md section 6.1)
_LATE_LOSS_READ = os.environ.get("MDM_HIP_EARLY_LOSS_SYNC", "0") != "1"


def _train_batch_fused(st, model, sample, optimizer, scheduler, logger, args, accumulate_gradient, num_grad_accumulations,
                       ema_model, loss_factor, lr, fp16):
    """The fused path of `train_batch`.  Same observable sequence as the reference (trainer.py:27-96), but the host never
    waits for the GPU in the middle of a step: the loss goes to pinned host memory by an asynchronous copy queued right
    behind the forward, backward + clip + AdamW + EMA are queued, and only then does the host look at the value -- by then
    the copy's event has long fired or fires while the GPU is busy with the backward.  What the reference decides on the
    host before backward ("NaN loss: drop the gradients, take no step") is decided on the device for the optimizer tail (a
    NaN loss poisons the gradient norm: no update, gradients cleared, step number kept) and replayed on the host afterwards
    for the rest (no EMA-counter increment, no logging, scheduler as the reference's branch has it)."""
    # several ranks in a synchronised step: a NaN loss on ONE rank must not change that rank's sequence of collectives --
    # every rank runs backward and the common tail; the update is skipped on the device by all of them (the REDUCED
    # gradient norm is not finite), and all of them learn it from that norm, so scheduler / EMA counter stay identical
    lockstep = st.reducer.world > 1 and not accumulate_gradient
    _settle(st)   # (a skipped previous step of a multi-rank run: see below)
    if ops._grad_sink is not st.reducer:
        st.activate()   # another FusedState (a second model in the same process) was used in between
    st.reducer.note_autocast(fp16)   # wire_dtype="auto": bf16 on the wire only for a bf16 step over RCCL
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=fp16):
        losses, times, x_t, means, targets, weights = model.get_loss(sample)
        loss = _loss_of(losses, weights)
        if fp16:
            loss = loss * loss_factor          # (the reference's fp32 branch has no loss_factor, trainer.py:58-63)
        if _LATE_LOSS_READ:
            st.post_scalar("loss", loss)
            loss_val = None
        else:
            loss_val = loss.item()
            if math.isnan(loss_val) and not lockstep:
                return _fused_nan_return(st, optimizer, scheduler, fp16, loss_val, losses, times, x_t, means, targets, True)
        if fp16 and num_grad_accumulations != 1:
            loss = loss / num_grad_accumulations
    st.reducer.mark("bw0")
    loss.backward()   # (fp32: the reference divides by num_grad_accumulations only AFTER this, trainer.py:73-75: no effect)
    # also after an accumulation micro-step: nothing queued (grouped weight gradients, GroupNorm parameter rows) may cross
    # into the next micro-step, where its ready() report would release a bucket before that step's own gradient is
    # written; inside no_sync() the reducer's finish() joins nothing
    st.finish_backward()
    if not accumulate_gradient:
        grp = optimizer.param_groups[0]
        decay = 0.0
        if ema_model is not None:   # ModelEma.update (models/model_ema.py:25-34), warm-up included
            decay = float(ema_model.counter >= ema_model.warmup_steps) * ema_model.decay
        st.optimizer_step(grp["lr"], grp["betas"], grp["eps"], grp["weight_decay"], float(getattr(args, "gradient_clip_norm", 2.0)),
                          decay, torch.bfloat16 if fp16 else torch.float32, loss=loss)
        optimizer._opt_called = True   # the step happened (silences lr_scheduler's call-order warning)
        skip_flag = st.post_skip_flag()
    if loss_val is None:
        loss_val = st.read_scalar("loss")
    if not lockstep and math.isnan(loss_val):
        # the device has skipped the update and cleared the arena (final micro-step), or the arena holds this micro-step's
        # NaNs on top of the earlier micro-steps' gradients, which the reference drops too (trainer.py:39, 66)
        return _fused_nan_return(st, optimizer, scheduler, fp16, loss_val, losses, times, x_t, means, targets, accumulate_gradient)
    if not accumulate_gradient:
        if ema_model is not None:
            ema_model.counter += 1
        if logger is not None:
            logger.add_scalar("train/Loss", loss_val)
            logger.add_scalar("lr", lr)
        scheduler.step()
    if not accumulate_gradient:
        # (One rank: the same flag covers what the loss value does not show -- an inf loss, or a finite loss whose gradients
        # overflowed: the device skips the update on the non-finite norm, and the host takes its bookkeeping back when it
        # sees the flag, exactly as below.  A NaN loss was dealt with above, on both sides.)
        # What a NaN anywhere did to this step is in the norm of the all-reduced gradient, identical on every rank -- but
        # it exists only when the whole step has run, and waiting for it here would cost the host its one-step lead over
        # the GPU (measured: +3...5 ms per step).  So the host-side consequences above (scheduler.step(), EMA counter) are
        # taken as if the step was applied, and corrected TWO calls later -- or when optimizer.state_dict() is taken -- if
        # the step turns out skipped (_settle): the reference's NaN branch (trainer.py:38-41: no scheduler step, no EMA
        # update in bf16), late by a fixed number of calls, on all ranks alike.  (Round 5 settled ONE call later: the
        # event of step N - 1 is recorded behind its optimizer, and at the start of call N the host has only waited for
        # forward N - 1 -- it blocked there for the whole previous backward + all-reduce + optimizer, the very wait this
        # scheme exists to avoid.  Two calls later the GPU is a whole step past the event.)
        st.__dict__.setdefault("unsettled", collections.deque()).append((skip_flag, scheduler, ema_model, fp16))
        if _STRICT_NAN:
            _settle(st, keep=0)
    return loss_val, losses, times, x_t, means, targets


# MDM_HIP_STRICT_NAN=1: a synchronised multi-rank step waits for its own skip flag before it returns (exact reference
# bookkeeping at every instant, at the price of the host's lead over the GPU)
_STRICT_NAN = os.environ.get("MDM_HIP_STRICT_NAN", "0") == "1"


def _settle(st, keep=1):
    """lock-step ranks: take back what the host did for synchronised steps that turned out skipped on the device (non-finite
    norm of the reduced gradient) as if they had been applied: the EMA warm-up counter, and -- bf16 branch, as the
    reference's NaN return -- the scheduler step.  The `keep` most recent steps stay unexamined: called with keep = 1 at the
    start of train_batch N it looks at step N - 2, whose flag arrived a whole step ago (no host stall; the same fixed delay
    on every rank, so learning rates never differ between ranks); keep = 0 (the optimizer's state_dict() hook,
    MDM_HIP_STRICT_NAN=1) waits for everything."""
    pend = getattr(st, "unsettled", None)
    while pend is not None and len(pend) > keep:
        (buf, ev), scheduler, ema_model, fp16 = pend.popleft()
        ev.synchronize()
        if math.isfinite(float(buf[0])):
            continue
        if ema_model is not None:
            ema_model.counter -= 1
        if fp16:
            # undo one scheduler.step(): schedulers whose rate is a function of last_epoch (LambdaLR and the closed-form ones)
            closed = getattr(scheduler, "_get_closed_form_lr", None)
            scheduler.last_epoch -= 1
            if hasattr(scheduler, "_step_count"):
                scheduler._step_count -= 1
            lrs = closed() if closed is not None else scheduler.get_lr()
            for g, lr in zip(scheduler.optimizer.param_groups, lrs):
                g["lr"] = lr
            scheduler._last_lr = [g["lr"] for g in scheduler.optimizer.param_groups]


def _fused_nan_return(st, optimizer, scheduler, fp16, loss_val, losses, times, x_t, means, targets, clear):
    """the reference's early return on a NaN loss (trainer.py:38-41 bf16: zero_grad, return; 64-69 fp32: zero_grad, a
    no-op optimizer.step(), scheduler.step(), return)"""
    if clear:
        _skip_step(st, optimizer, None, False, None, fp16)
    if not fp16:
        scheduler.step()
    return loss_val, losses, times, x_t, means, targets



def _skip_step(st, optimizer, loss, accumulate_gradient, args, fp16):
    """NaN loss on the fused path: drop the gradients accumulated so far (reference trainer.py:39, 66).  Only reached where
    no other rank depends on this one's sequence (one rank, or an accumulation micro-step under no_sync()); in a
    synchronised multi-rank step train_batch does NOT return before the collectives -- with the reference's per-rank early
    return the other ranks wait for bucket all-reduces that never come."""
    st.finish_backward()    # nothing may still be adding into the arena
    st.reducer.zero_grad()


class ModelEma(torch.nn.Module):
    """``ml_mdm.models.model_ema.ModelEma`` (reference models/model_ema.py:11-52): a deep copy of the vision model whose
    state follows ``ema = decay * ema + (1 - decay) * model``.  Provided so the package is self-contained; the
    reference's own class works with ``train_batch`` just as well (same attributes)."""

    def __init__(self, model, decay=0.9999, warmup_steps=0, device=None):
        super().__init__()
        from copy import deepcopy

        self.module = deepcopy(model)
        self.module.eval()
        self.decay, self.device, self.warmup_steps, self.counter = decay, device, warmup_steps, 0
        if device is not None:
            self.module.to(device=device)

    def update(self, model):
        decay = (self.counter >= self.warmup_steps) * self.decay
        self.counter += 1
        with torch.no_grad():
            msd = model.state_dict()
            for k, ema_v in self.module.state_dict().items():
                mv = msd[k].detach()
                if self.device:
                    mv = mv.to(device=self.device)
                ema_v.mul_(decay).add_(mv, alpha=1.0 - decay)

    def save(self, fname, other_items=None):
        ckpt = {"state_dict": self.module.state_dict()}
        ckpt.update(other_items or {})
        torch.save(ckpt, fname)

    def load(self, fname):
        ckpt = torch.load(fname, map_location="cpu")
        mine = self.module.state_dict()
        self.module.load_state_dict({k: v for k, v in ckpt["state_dict"].items() if k in mine}, strict=False)


# --------------------------------------------------------------------------------------------------------------
# self-contained step object
# --------------------------------------------------------------------------------------------------------------
class TrainStep:
    """fused=True (GPU): the arenas and the fused tail of ``FusedState`` with a fixed learning rate; one optimizer step
    per call.  fused=False keeps torch's optimizer on the flat gradient arena (CPU tests, or as a cross-check)."""

    def __init__(self, pipeline, lr=5e-5, clip_norm=2.0, ema_decay=0.9999, bf16=True, use_ema=True,
                 bucket_mb=256.0, wire_dtype=None, fused=None, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 async_wgrad=True):
        self.pipeline = pipeline
        self.net = pipeline.get_model().vision_model
        self.params = [p for p in self.net.parameters() if p.requires_grad]
        self.fused = self.params[0].is_cuda if fused is None else fused
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.clip_norm, self.bf16, self.ema_decay = clip_norm, bf16, ema_decay
        self.steps = 0
        if self.fused:
            pipeline.materialize_targets = False   # the fused loss kernel never forms the target tensor; nobody reads it here
            self.state = FusedState(self.net, bucket_mb=bucket_mb, wire_dtype=wire_dtype, use_ema=use_ema, async_wgrad=async_wgrad)
            self.reducer = self.state.reducer
            self.opt = None
            self.ema = None
        else:
            self.reducer = GradReducer(self.params, bucket_mb=bucket_mb, wire_dtype=wire_dtype)
            self.reducer.broadcast_parameters(0)
            self.opt = torch.optim.AdamW(self.params, lr=lr, betas=betas, weight_decay=weight_decay, eps=eps)
            self.ema = [p.detach().clone() for p in self.params] if use_ema else None

    # the arenas under their historical names
    flat_p = property(lambda self: self.state.flat_p)
    m = property(lambda self: self.state.m)
    v = property(lambda self: self.state.v)
    flat_ema = property(lambda self: self.state.flat_ema)
    step_dev = property(lambda self: self.state.step_dev)

    def ema_state(self):
        """name -> EMA tensor (views of the flat EMA arena in fused mode)"""
        names = {id(p): n for n, p in self.net.named_parameters()}
        if self.fused:
            return {names[id(p)]: self.state.view(self.state.flat_ema, p) for p in self.params}
        return {names[id(p)]: e for p, e in zip(self.params, self.ema)}

    def __call__(self, sample, **loss_kw):
        self.pipeline.train()
        dev_type = "cuda" if self.params[0].is_cuda else "cpu"
        with torch.autocast(dev_type, dtype=torch.bfloat16, enabled=self.bf16):
            losses, times, x_t, means, targets, weights = self.pipeline.get_loss(sample, **loss_kw)
            loss = _loss_of(losses, weights)
        if self.fused:
            # Nothing branches on the loss before backward: a NaN / inf loss gives a non-finite gradient norm, and the fused
            # optimizer then skips the update, clears the gradients and does not advance its (device-side) step number --
            # same outcome as the reference's early return (trainer.py:38-41), identical on every rank without a
            # collective.  The value itself travels to pinned host memory behind the forward and is read once backward
            # and the optimizer tail are queued: the host never waits for the GPU in the middle of a step.
            self.state.reducer.note_autocast(self.bf16)
            self.state.post_scalar("loss", loss)
            loss.backward()
            self.state.finish_backward()
            self.state.optimizer_step(self.lr, self.betas, self.eps, self.weight_decay, self.clip_norm, self.ema_decay,
                                      torch.bfloat16 if self.bf16 else torch.float32, loss=loss)
            loss_val = self.state.read_scalar("loss")
            self.steps += 1 if math.isfinite(loss_val) else 0
            return loss_val
        loss_val = loss.item()  # the reference syncs here every step (trainer.py:37)
        bad = not math.isfinite(loss_val)
        if self.reducer.world > 1:
            # every rank must take the same branch, or the ranks that run backward wait for bucket all-reduces the
            # skipping rank never issues (the reference's DDP hangs the same way: trainer.py:38-41 returns per rank)
            flag = torch.tensor([1.0 if bad else 0.0], device=self.params[0].device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX, group=self.reducer.group)
            bad = bool(flag.item() > 0)
        if bad:
            self.reducer.zero_grad()
            return loss_val
        loss.backward()
        ops.flush_wgrad_queue()
        ops.join_side_stream()
        self.reducer.finish()
        self.steps += 1
        gnorm = torch.linalg.vector_norm(self.reducer.flat)
        scale = torch.clamp(self.clip_norm / (gnorm + 1e-6), max=1.0)
        self.reducer.flat.mul_(scale)  # == clip_grad_norm_ over all parameters (one pass over the arena)
        self.opt.step()
        ops.invalidate_packed_weights()
        if self.ema is not None:
            with torch.no_grad():
                torch._foreach_lerp_(self.ema, [p.detach() for p in self.params], 1.0 - self.ema_decay)
        self.reducer.zero_grad()
        return loss_val

"""The train step of the headline metric: forward (+bf16 autocast) -> loss -> backward ->
gradient all-reduce -> clip -> AdamW -> EMA -> zero-grad.

Mirrors ``ml_mdm.trainer.train_batch`` (reference trainer.py:13-96), the optimizer set-up of
``clis/train_parallel.py:122-134`` (AdamW, weight_decay 0, eps 1e-8) and ``ModelEma.update``
(models/model_ema.py:25-34; the EMA here tracks parameters -- the model has no persistent
buffers).  bf16 needs no loss scaling, so the reference's GradScaler is not reproduced.
On the GPU the optimizer tail is fused: gradients land directly in a flat arena
and ``mdm_sumsq`` + ``mdm_adamw_ema_step`` do clip + AdamW + EMA + zero-grad in two streaming passes.
"""
import math

import torch

from . import ops
from .distributed import GradReducer


class TrainStep:
    """fused=True (GPU): parameters, gradients, Adam moments and the EMA live in flat fp32 arenas; backward kernels
    add parameter gradients straight into the gradient arena (ops.set_grad_sink) and the whole optimizer tail is
    ``mdm_sumsq`` + ``mdm_adamw_ema_step`` (clip, AdamW, EMA, zero-grad in one pass).  fused=False keeps torch's
    optimizer (CPU tests, or as a cross-check)."""

    def __init__(self, pipeline, lr=5e-5, clip_norm=2.0, ema_decay=0.9999, bf16=True, use_ema=True,
                 bucket_mb=256.0, wire_dtype=None, fused=None, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 async_wgrad=True):
        self.pipeline = pipeline
        self.net = pipeline.get_model().vision_model
        self.params = [p for p in self.net.parameters() if p.requires_grad]
        self.reducer = GradReducer(self.params, bucket_mb=bucket_mb, wire_dtype=wire_dtype)
        self.reducer.broadcast_parameters(0)
        self.fused = self.params[0].is_cuda if fused is None else fused
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.clip_norm, self.bf16, self.ema_decay = clip_norm, bf16, ema_decay
        self.steps = 0
        if self.fused:
            pipeline.materialize_targets = False   # the fused loss kernel never forms the target tensor; nobody reads it here
        if self.fused:
            # flat parameter arena in the SAME order as the gradient arena (reverse registration order)
            flat = torch.empty_like(self.reducer.flat)
            off = 0
            with torch.no_grad():
                for p in reversed(self.params):
                    n = p.numel()
                    flat[off:off + n].copy_(p.reshape(-1))
                    p.data = flat[off:off + n].view_as(p)
                    off += n
            self.flat_p = flat
            self.m = torch.zeros_like(flat)
            self.v = torch.zeros_like(flat)
            self.flat_ema = flat.clone() if use_ema else None
            self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=flat.device)
            self.step_dev = torch.zeros(1, dtype=torch.int32, device=flat.device)   # optimizer step number, on the device
            self.reducer.rebind()
            ops.set_grad_sink(self.reducer)
            ops.enable_async_wgrad(async_wgrad)
            ops.enable_deferred_wgrad(async_wgrad, max_group=32 if self.reducer.world == 1 else 13)
            ops.invalidate_packed_weights()
            self.opt = None
            self.ema = None
        else:
            self.opt = torch.optim.AdamW(self.params, lr=lr, betas=betas, weight_decay=weight_decay, eps=eps)
            self.ema = [p.detach().clone() for p in self.params] if use_ema else None

    def ema_state(self):
        """name -> EMA tensor (views of the flat EMA arena in fused mode)"""
        names = {id(p): n for n, p in self.net.named_parameters()}
        if self.fused:
            out, off = {}, 0
            for p in reversed(self.params):
                out[names[id(p)]] = self.flat_ema[off:off + p.numel()].view_as(p)
                off += p.numel()
            return out
        return {names[id(p)]: e for p, e in zip(self.params, self.ema)}

    def __call__(self, sample, **loss_kw):
        self.pipeline.train()
        dev_type = "cuda" if self.params[0].is_cuda else "cpu"
        with torch.autocast(dev_type, dtype=torch.bfloat16, enabled=self.bf16):
            losses, times, x_t, means, targets, weights = self.pipeline.get_loss(sample, **loss_kw)
            loss = losses.mean() if weights is None else (losses * weights).sum() / weights.sum()
        if self.fused:
            # The loss is read where the reference reads it (trainer.py:37) -- by then the host has queued the whole
            # forward, and it queues the next step's forward while this step's backward runs -- but nothing branches on
            # it: a NaN / inf loss gives a non-finite gradient norm, and the fused optimizer then skips the update,
            # clears the gradients and does not advance its (device-side) step number.  Same outcome as the reference's
            # early return (trainer.py:38-41), identical on every rank without a collective.
            loss_val = loss.item()
            loss.backward()
            ops.flush_wgrad_queue()
            ops.join_side_stream()
            self.reducer.finish()
            ops.sumsq(self.reducer.flat, out=self.gnorm_sq, step_counter=self.step_dev)
            ops.adamw_ema_step(self.flat_p, self.reducer.flat, self.m, self.v, self.flat_ema, self.gnorm_sq, self.lr,
                               self.betas[0], self.betas[1], self.eps, self.weight_decay, 0, self.clip_norm,
                               self.ema_decay, zero_grad=True, step_dev=self.step_dev)
            ops.repack_all(torch.bfloat16 if self.bf16 else torch.float32)   # one launch instead of one per weight
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

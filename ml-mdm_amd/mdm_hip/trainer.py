"""The train step of the headline metric: forward (+bf16 autocast) -> loss -> backward ->
gradient all-reduce -> clip -> AdamW -> EMA -> zero-grad.

Mirrors ``ml_mdm.trainer.train_batch`` (reference trainer.py:13-96), the optimizer set-up of
``clis/train_parallel.py:122-134`` (AdamW, weight_decay 0, eps 1e-8) and ``ModelEma.update``
(models/model_ema.py:25-34; the EMA here tracks parameters -- the model has no persistent
buffers).  bf16 needs no loss scaling, so the reference's GradScaler is not reproduced.
The optimizer tail currently runs as torch multi-tensor ops over the flat gradient arena
(fusing clip + AdamW + EMA + weight re-pack into one HIP pass
is the next step).
"""
import torch

from .distributed import GradReducer


class TrainStep:
    def __init__(self, pipeline, lr=5e-5, clip_norm=2.0, ema_decay=0.9999, bf16=True, use_ema=True,
                 bucket_mb=256.0, wire_dtype=None):
        self.pipeline = pipeline
        self.net = pipeline.get_model().vision_model
        self.params = [p for p in self.net.parameters() if p.requires_grad]
        self.reducer = GradReducer(self.params, bucket_mb=bucket_mb, wire_dtype=wire_dtype)
        self.reducer.broadcast_parameters(0)
        self.opt = torch.optim.AdamW(self.params, lr=lr, weight_decay=0, eps=1e-8, fused=self.params[0].is_cuda)
        self.clip_norm, self.bf16 = clip_norm, bf16
        self.ema_decay = ema_decay
        self.ema = [p.detach().clone() for p in self.params] if use_ema else None
        self.steps = 0

    def __call__(self, sample, **loss_kw):
        self.pipeline.train()
        dev_type = "cuda" if self.params[0].is_cuda else "cpu"
        with torch.autocast(dev_type, dtype=torch.bfloat16, enabled=self.bf16):
            losses, times, x_t, means, targets, weights = self.pipeline.get_loss(sample, **loss_kw)
            loss = losses.mean() if weights is None else (losses * weights).sum() / weights.sum()
        loss_val = loss.item()  # the reference syncs here every step (trainer.py:37)
        if loss_val != loss_val:
            self.reducer.zero_grad()
            return loss_val
        loss.backward()
        self.reducer.finish()
        gnorm = torch.linalg.vector_norm(self.reducer.flat)
        scale = torch.clamp(self.clip_norm / (gnorm + 1e-6), max=1.0)
        self.reducer.flat.mul_(scale)  # == clip_grad_norm_ over all parameters (one pass over the arena)
        self.opt.step()
        if self.ema is not None:
            with torch.no_grad():
                torch._foreach_lerp_(self.ema, [p.detach() for p in self.params], 1.0 - self.ema_decay)
        self.reducer.zero_grad()
        self.steps += 1
        return loss_val

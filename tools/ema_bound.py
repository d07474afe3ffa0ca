#!/usr/bin/env python
"""Upper bound of moving the EMA update off the critical path: the UNet-64 train step with and without an EMA model
(the fused optimizer kernel then streams 32 instead of 40 bytes per parameter).   gpurun -- python tools/ema_bound.py"""
import gc
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from mdm_hip import trainer  # noqa: E402


def run(with_ema, steps=12):
    dev = torch.device("cuda:0")
    pipe, side = bench.build("unet64", dev)
    vm = pipe.model.vision_model
    opt = torch.optim.AdamW(vm.parameters(), lr=5e-5, weight_decay=0, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 1.0)
    ema = trainer.ModelEma(vm) if with_ema else None
    args = types.SimpleNamespace(fp16=True, gradient_clip_norm=2.0)
    scaler = torch.amp.GradScaler("cuda")
    sample = bench.synthetic_batch(64, side, dev, seed=1)

    def step():
        return trainer.train_batch(pipe, sample, opt, sched, None, args, grad_scaler=scaler, accumulate_gradient=False,
                                   num_grad_accumulations=1, ema_model=ema, loss_factor=1.0)[0]
    for _ in range(12):
        step()
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    gc.enable()
    del pipe, opt, ema
    torch.cuda.empty_cache()
    return dt


if __name__ == "__main__":
    for _ in range(2):
        print("with EMA %.2f ms   without %.2f ms" % (run(True), run(False)), flush=True)

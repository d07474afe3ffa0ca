#!/usr/bin/env python
"""How far the host runs ahead of the GPU inside a train step: at every flush point of backward (resolution-level
boundaries, the end of backward) and at the step's ends, the host's time of ENQUEUEING an event against the time the GPU
reaches it.  lead = GPU time - host time (ms): what the host can spend without the GPU noticing.
   gpurun -- python tools/host_lead.py [unet64|nested256] [--force-collectives]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from mdm_hip import ops  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "unet64"
    force = "--force-collectives" in sys.argv
    dev = torch.device("cuda:0")
    if force:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0)
    pipe, side = bench.build(workload, dev)
    step, opt = bench.make_step(pipe, True, 1, bucket_mb=64.0, force_collectives=force)
    sample = bench.synthetic_batch(64 if workload == "unet64" else 16, side, dev, seed=1)
    for _ in range(8):
        step(sample)
    marks = []
    real_flush = ops.flush_wgrad_queue

    def mark(tag):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append((tag, time.perf_counter(), ev))

    def flush():
        real_flush()
        mark("flush point")

    ops.flush_wgrad_queue = flush
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    base = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    base.record()
    for i in range(3):
        mark("step %d begins (host enters train_batch)" % i)
        step(sample)
        mark("step %d: train_batch returned" % i)
    torch.cuda.synchronize()
    print("%-46s %10s %10s %8s" % ("", "host ms", "GPU ms", "lead"))
    for tag, th, ev in marks:
        tg = base.elapsed_time(ev)
        print("%-46s %10.2f %10.2f %8.2f" % (tag, (th - t0) * 1e3, tg, tg - (th - t0) * 1e3))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""How close the train step is to host-bound: wall time per step against the CPU time the process spends per step
(all threads: the Python thread + the autograd engine's) -- the host is the limit where they meet.
   gpurun -- python tools/host_lead.py [unet64|nested256] [steps]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "nested256"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    pipe, side = bench.build(workload, dev)
    step, opt = bench.make_step(pipe, True, 1)
    sample = bench.synthetic_batch(64 if workload == "unet64" else 16, side, dev, seed=1)
    for _ in range(8):
        step(sample)
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    w0, c0 = time.perf_counter(), time.process_time()
    for _ in range(steps):
        step(sample)
    c1 = time.process_time()
    torch.cuda.synchronize()
    w1 = time.perf_counter()
    print("%s: wall %.2f ms per step, process CPU time %.2f ms per step (%d steps)" % (workload, (w1 - w0) / steps * 1e3, (c1 - c0) / steps * 1e3, steps))


if __name__ == "__main__":
    main()

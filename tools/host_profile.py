#!/usr/bin/env python
"""Where the HOST spends a train step (cProfile, backward run in the calling thread so that its Python frames are seen):
   gpurun -- python tools/host_profile.py [unet64|nested256] [steps]"""
import cProfile
import gc
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "nested256"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    pipe, side = bench.build(workload, dev)
    step, opt = bench.make_step(pipe, True, 1)
    sample = bench.synthetic_batch(64 if workload == "unet64" else 16, side, dev, seed=1)
    torch.autograd.set_multithreading_enabled(False)
    for _ in range(6):
        step(sample)
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step(sample)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.strip_dirs()
    print("== by own time (per %d steps)" % steps)
    st.sort_stats("tottime").print_stats(45)
    print("== by cumulative time")
    st.sort_stats("cumulative").print_stats(60)


if __name__ == "__main__":
    main()

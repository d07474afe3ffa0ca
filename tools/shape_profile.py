#!/usr/bin/env python
"""Per-shape table of the GEMM-class launches of ONE train step (HIP events on the launch stream, in-step).
   gpurun -- python tools/shape_profile.py [unet64|nested256] [--serial]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from mdm_hip import ops  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "unet64"
    dev = torch.device("cuda:0")
    pipe, side = bench.build(workload, dev)
    step, opt = bench.make_step(pipe, True, 1, serial_wgrad="--serial" in sys.argv)
    sample = bench.synthetic_batch(64 if workload == "unet64" else 16, side, dev, seed=1)
    for _ in range(3):
        step(sample)
    torch.cuda.synchronize()
    ops.profile_begin(shapes=True)
    step(sample)
    torch.cuda.synchronize()
    roof = ops.profile_end(bench.PEAK_BF16_TFLOPS)
    rows = sorted(roof["all_gemm_kernels"].items(), key=lambda kv: -kv[1]["time_ms"])
    tot = sum(v["time_ms"] for _, v in rows)
    print("GEMM-class launches of one %s step: %.2f ms, %.1f TF/s FLOP-weighted" % (workload, tot, roof["gemm_weighted"]["tflops"]))
    for k, v in rows[:45]:
        print("%8.3f ms  x%-3d %7.1f TF  %s" % (v["time_ms"], v["launches"], v["tflops"], k))
    print("HBM-class:")
    for k, v in roof["hbm_kernels"].items():
        print("%8.3f ms  x%-3d %7.0f GB/s  %s" % (v["time_ms"], v["launches"], v["gb_per_s"], k))


if __name__ == "__main__":
    main()

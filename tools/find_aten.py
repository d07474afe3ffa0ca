#!/usr/bin/env python
"""Which Python lines launch the non-mdm GPU work of a train step (ATen elementwise kernels, fills, device copies)?
torch.profiler over one step of bench.py's loop, grouped by the innermost repo frame.  Development tool:
   gpurun -- python tools/find_aten.py [unet64|nested256]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "unet64"
    dev = torch.device("cuda:0")
    pipe, side = bench.build(workload, dev)
    step, opt = bench.make_step(pipe, True, 1)
    sample = bench.synthetic_batch(64 if workload == "unet64" else 16, side, dev, seed=1)
    for _ in range(3):
        step(sample)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step(sample)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0, set()])
    for ev in prof.events():
        if ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
            continue
        if ev.cpu_children:   # count leaves only (aten::add -> kernel), not their aten:: parents
            if any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
                continue
        frame = next((f for f in ev.stack if "/mdm_hip/" in f or "bench.py" in f), (ev.stack[0] if ev.stack else "?"))
        key = (ev.name, frame.strip()[-110:])
        a = agg[key]
        a[0] += 1
        a[1] += ev.device_time_total
        a[2].add(str(ev.input_shapes)[:80])
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    print("ATen device work in one train step (%s): %.3f ms total" % (workload, sum(v[1] for v in agg.values()) / 1e3))
    for (name, frame), (n, t, shapes) in rows[:40]:
        print("%8.1f us  x%-4d %-28s %s   %s" % (t, n, name, frame, sorted(shapes)[:2]))


if __name__ == "__main__":
    main()

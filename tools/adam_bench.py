#!/usr/bin/env python
"""mdm_adamw_ema_step on flat arenas of the UNet-64's size (415 M parameters), HIP events:
   gpurun -- python tools/adam_bench.py        (MDM_HIP_LIB selects a variant library)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402

from mdm_hip import _lib  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 461_000_000 // 4 * 4
    dev = torch.device("cuda:0")
    skew = int(os.environ.get("ADAM_SKEW_BYTES", "0")) // 4   # development: shift array k by k * skew bytes (channel / bank alignment)
    bufs = [torch.randn(n + 5 * skew, device=dev) * 0.01 for _ in range(5)]
    p, g, m, v, e = (b[k * skew:k * skew + n] for k, b in enumerate(bufs))
    print("base addresses mod 1 MiB:", [hex(t.data_ptr() % (1 << 20)) for t in (p, g, m, v, e)])
    v.abs_()
    q = torch.ones(1, device=dev)
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream

    def one(i):
        _lib.check(L.mdm_adamw_ema_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), e.data_ptr(), q.data_ptr(), n, 1e-4, 0.9, 0.999,
                                        1e-8, 0.01, i + 1, None, 2.0, 0.9999, 1, st), "adamw")
    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10):
        one(i + 3)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print("adamw_ema lib=%s n=%d  %.3f ms  %.0f GB/s" % (os.path.basename(os.environ.get("MDM_HIP_LIB", "product")), n, t, 40.0 * n / t / 1e6))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""What the vendor GEMM (hipBLASLt through torch.matmul, bf16) reaches on the GEMM shapes of the cc12m_64x64 U-Net:
the yardstick for the hand-written implicit-GEMM kernels (which additionally gather 3x3 taps and fuse epilogues).
   gpurun -- python tools/blas_ref.py"""
import torch

SHAPES = [("3x3 256->256 @64 (M=262144 K=2304 N=256)", 262144, 2304, 256),
          ("3x3 512->512 @32 (M=65536 K=4608 N=512)", 65536, 4608, 512),
          ("3x3 768->768 @16 (M=16384 K=6912 N=768)", 16384, 6912, 768),
          ("1x1 768->3072 @16", 16384, 768, 3072), ("1x1 3072->768 @16", 16384, 3072, 768),
          ("1x1 768->2304 @16", 16384, 768, 2304), ("1x1 768->768 @16", 16384, 768, 768),
          ("1x1 512->2048 @32", 65536, 512, 2048), ("square 8192", 8192, 8192, 8192)]


def main():
    dev = torch.device("cuda:0")
    for name, M, K, N in SHAPES:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = torch.randn(N, K, device=dev).to(torch.bfloat16)
        for _ in range(3):
            torch.matmul(a, b.t())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.matmul(a, b.t())
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10e3
        print("%-44s %8.3f ms  %7.1f TF/s" % (name, t * 1e3, 2.0 * M * K * N / t / 1e12), flush=True)


if __name__ == "__main__":
    main()

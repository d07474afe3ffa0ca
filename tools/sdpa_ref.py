#!/usr/bin/env python
"""What torch's scaled_dot_product_attention (the flash kernels shipped with PyTorch-ROCm) reaches on the two attention
shapes of the cc12m_64x64 U-Net -- self-attention only, [B, H, L, d] contiguous -- as a yardstick for attention.hip
(which also does the 32 text keys of every layer and reads the strided qkv projection in place).
   gpurun -- python tools/sdpa_ref.py"""
import torch
import torch.nn.functional as F


def main():
    dev = torch.device("cuda:0")
    for L, d in ((1024, 64), (256, 96)):
        B, H = 64, 8
        q, k, v = [torch.randn(B, H, L, d, device=dev).to(torch.bfloat16).requires_grad_() for _ in range(3)]
        fl = 4.0 * B * H * L * L * d
        for backend in ("flash", "efficient", "math"):
            try:
                from torch.nn.attention import SDPBackend, sdpa_kernel
                be = {"flash": SDPBackend.FLASH_ATTENTION, "efficient": SDPBackend.EFFICIENT_ATTENTION, "math": SDPBackend.MATH}[backend]
                with sdpa_kernel(be):
                    o = F.scaled_dot_product_attention(q, k, v)
                    go = torch.randn_like(o)
                    for _ in range(2):
                        F.scaled_dot_product_attention(q, k, v)
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    e[0].record()
                    for _ in range(10):
                        F.scaled_dot_product_attention(q.detach(), k.detach(), v.detach())
                    e[1].record()
                    for _ in range(10):
                        torch.autograd.grad(o, (q, k, v), go, retain_graph=True)
                    e[2].record()
                    torch.cuda.synchronize()
                    tf, tb = e[0].elapsed_time(e[1]) / 10e3, e[1].elapsed_time(e[2]) / 10e3
                    print("sdpa %-9s L=%d d=%d  fwd %7.3f ms %6.1f TF   bwd %7.3f ms %6.1f TF" % (backend, L, d, tf * 1e3, fl / tf / 1e12, tb * 1e3, 2.5 * fl / tb / 1e12), flush=True)
            except Exception as ex:  # a backend that is not built for this shape / platform
                print("sdpa %-9s L=%d d=%d  unavailable: %s" % (backend, L, d, str(ex)[:80]), flush=True)


if __name__ == "__main__":
    main()

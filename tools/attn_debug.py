#!/usr/bin/env python
"""Development aid for the attention kernels: the forward kernel and
every backward path (mdm_dev_set_attn_bwd 1 = split, 2 = small
(32x32 MFMA where the shape allows), 3 = small16) against a torch fp32 reference on the same bf16 inputs, error per output
tensor, and -- when one is off -- a map of the worst 32 x 32 blocks of the first failing (batch, head).
   gpurun -- python tools/attn_debug.py            # correctness cases
   gpurun -- python tools/attn_debug.py time       # + timings at the U-Net's shapes (batch 64)"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402

from mdm_hip import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
MODES = {1: "split16", 2: "small32", 3: "small16", 4: "stream32"}


def ref_attn(qkv, kvc, mask, H):
    B, L, C3 = qkv.shape
    C = C3 // 3
    d = C // H
    q, k, v = [t.reshape(B, L, H, d).transpose(1, 2) for t in qkv.split(C, -1)]
    sc = 1.0 / math.sqrt(d)
    out = torch.softmax(q @ k.transpose(-1, -2) * sc, -1) @ v
    if kvc is not None:
        S = kvc.shape[1]
        kc, vc = [t.reshape(B, S, H, d).transpose(1, 2) for t in kvc.split(C, -1)]
        sx = q @ kc.transpose(-1, -2) * sc
        if mask is not None:
            sx = sx.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
        out = out + torch.softmax(sx, -1) @ vc
    return out.transpose(1, 2).reshape(B, L, C)


def relerr(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


def block_map(name, got, want, H, d, rows):
    """worst blocks of head 0.. of batch 0: got / want [B, rows, parts * H * d]"""
    B = got.shape[0]
    parts = got.shape[-1] // (H * d)
    e = (got - want).abs()
    scale = float(want.abs().max())
    worst = []
    for b in range(B):
        for part in range(parts):
            for h in range(H):
                blk = e[b, :, part * H * d + h * d: part * H * d + (h + 1) * d]
                worst.append((float(blk.max()) / scale, b, part, h))
    worst.sort(reverse=True)
    print("    %s: worst (err/max, batch, part, head): %s" % (name, ", ".join("%.3f b%d p%d h%d" % w for w in worst[:4])))
    _, b, part, h = worst[0]
    blk = e[b, :, part * H * d + h * d: part * H * d + (h + 1) * d] / scale
    for r0 in range(0, rows, 32):
        print("      rows %3d-%3d: " % (r0, min(rows, r0 + 32) - 1) +
              "  ".join("%.3f" % float(blk[r0:r0 + 32, c0:c0 + 32].max()) for c0 in range(0, d, 32)))
    # finer: within the worst 32-row block, per row and per 4-column group
    r0 = int(blk.max(dim=1).values.argmax()) // 32 * 32
    sub = blk[r0:r0 + 32]
    print("      rows of block %d (max over columns): %s" % (r0, " ".join("%.2f" % float(x) for x in sub.max(dim=1).values)))
    print("      columns (max over those rows):       %s" % " ".join("%.2f" % float(x) for x in sub.max(dim=0).values))


def case(B, L, S, H, d, masked, seed=6):
    g = torch.Generator().manual_seed(seed)
    C = H * d
    qkv = (torch.randn(B, L, 3 * C, generator=g) * 1.2).bfloat16().float().requires_grad_()
    kvc = (torch.randn(B, S, 2 * C, generator=g) * 1.2).bfloat16().float().requires_grad_() if S else None
    mask = None
    if masked and S:
        mask = torch.ones(B, S)
        mask[0, S // 2:] = 0
        mask[-1, 1:3] = 0
    o_ref = ref_attn(qkv, kvc, mask, H)
    go = torch.randn(o_ref.shape, generator=g).bfloat16().float()
    o_ref.backward(go)
    ok = True
    for fname in ("fwd16",):      # the forward kernel (the 32x32x16 forward of round 5 was removed in round 6)
        qd = qkv.detach().bfloat16().to(dev)
        kd = kvc.detach().bfloat16().to(dev) if S else None
        o = ops.attention(qd, kd, mask.to(dev) if mask is not None else None, H).float().cpu()
        e = relerr(o, o_ref.detach())
        print("B=%d L=%d S=%d H=%d d=%d masked=%d  %-8s out %.4f  %s" % (B, L, S, H, d, masked, fname, e, "ok" if e < 3e-2 else "FAIL"), flush=True)
        if not e < 3e-2:
            ok = False
            block_map("out", o, o_ref.detach(), H, d, L)
    for mode, name in MODES.items():
        _lib.lib().mdm_dev_set_attn_bwd(mode)
        qd = qkv.detach().bfloat16().to(dev).requires_grad_()
        kd = kvc.detach().bfloat16().to(dev).requires_grad_() if S else None
        md = mask.to(dev) if mask is not None else None
        o = ops.attention(qd, kd, md, H)
        o.backward(go.bfloat16().to(dev))
        torch.cuda.synchronize()
        gq = qd.grad.float().cpu()
        errs = {"dq": relerr(gq[..., :C], qkv.grad[..., :C]), "dk": relerr(gq[..., C:2 * C], qkv.grad[..., C:2 * C]),
                "dv": relerr(gq[..., 2 * C:], qkv.grad[..., 2 * C:])}
        if S:
            gk = kd.grad.float().cpu()
            errs["dkc"] = relerr(gk[..., :C], kvc.grad[..., :C])
            errs["dvc"] = relerr(gk[..., C:], kvc.grad[..., C:])
        bad = [k for k, v in errs.items() if not v < 3e-2]
        print("B=%d L=%d S=%d H=%d d=%d masked=%d  %-8s %s  %s" % (B, L, S, H, d, masked, name,
              " ".join("%s %.4f" % kv for kv in errs.items()), "FAIL " + ",".join(bad) if bad else "ok"), flush=True)
        if bad:
            ok = False
            block_map("dqkv", gq, qkv.grad, H, d, L)
            if S and ("dkc" in bad or "dvc" in bad):
                block_map("dkvc", gk, kvc.grad, H, d, S)
    _lib.lib().mdm_dev_set_attn_bwd(0)
    return ok


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def timing(B=64):
    for L, d in ((256, 96), (1024, 64), (256, 64)):
        C = 8 * d
        qkv = torch.randn(B, L, 3 * C, device=dev).bfloat16().requires_grad_()
        kvc = torch.randn(B, 32, 2 * C, device=dev).bfloat16().requires_grad_()
        fl = 4.0 * B * 8 * L * (L + 32) * d
        line = "attn B=%d L=%d d=%d " % (B, L, d)
        for fname in ("fwd16",):
            tf = timeit(lambda: ops.attention(qkv.detach(), kvc.detach(), None, 8))
            line += " %s %.3f ms %.0f TF" % (fname, tf, fl / tf / 1e9)
        line += " |"
        for mode, name in MODES.items():
            _lib.lib().mdm_dev_set_attn_bwd(mode)
            o = ops.attention(qkv, kvc, None, 8)
            go = torch.randn_like(o)
            tb = timeit(lambda: torch.autograd.grad(o, (qkv, kvc), go, retain_graph=True))
            line += "  bwd[%s] %.3f ms %.0f TF" % (name, tb, 2.5 * fl / tb / 1e9)
        _lib.lib().mdm_dev_set_attn_bwd(0)
        print(line, flush=True)


def stamps(B=64, L=256, d=96):
    """phase time stamps of attn_bwd_small32_kernel (mdm_dev_set_attn_dbg): shader cycles between the stamps, mean over blocks"""
    C = 8 * d
    qkv = torch.randn(B, L, 3 * C, device=dev).bfloat16().requires_grad_()
    kvc = torch.randn(B, 32, 2 * C, device=dev).bfloat16().requires_grad_()
    _lib.lib().mdm_dev_set_attn_bwd(2)
    o = ops.attention(qkv, kvc, None, 8)
    go = torch.randn_like(o)
    torch.autograd.grad(o, (qkv, kvc), go, retain_graph=True)
    buf = torch.zeros(B * 8, 8, 16, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    import ctypes
    _lib.lib().mdm_dev_set_attn_dbg(ctypes.c_void_p(buf.data_ptr()))
    torch.autograd.grad(o, (qkv, kvc), go, retain_graph=True)
    torch.cuda.synchronize()
    _lib.lib().mdm_dev_set_attn_dbg(None)
    _lib.lib().mdm_dev_set_attn_bwd(0)
    t = buf.cpu().double()
    names = ["start", "K committed", "Q-phase operands + delta", "barrier + tile 0 operands", "Q loop", "dq stored", "barrier",
             "dO staged + barrier", "K self loop", "dk / dv stored", "text step", "text reduce", "text stored"]
    t0 = t[:, :, 0:1].min()
    print("stamps B=%d L=%d d=%d: cycles since the previous stamp (mean over blocks; wave 0 | wave 7), start spread %.0f" %
          (B, L, d, float(t[:, 0, 0].max() - t0)))
    prev = t[:, :, 0]
    for i in range(1, 13):
        cur = t[:, :, i]
        ok = cur > 0
        dlt = (cur - prev)
        print("   %-28s  %9.0f | %9.0f    (max %9.0f)" % (names[i], float(dlt[:, 0].mean()), float(dlt[:, 7].mean()), float(dlt.max())))
        prev = torch.where(ok, cur, prev)
    tot = t[:, :, 12] - t[:, :, 0]
    print("   block total: mean %.0f cycles, max %.0f; first block start -> last block end %.0f" %
          (float(tot.mean()), float(tot.max()), float(t[:, :, 12].max() - t0)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "timeonly":
        timing(int(os.environ.get("KB_BATCH", "64")))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stamps":
        stamps()
        stamps(64, 256, 64)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pmc":      # a few launches of every backward path, for rocprofv3 --pmc
        for L, d in ((256, 96), (1024, 64)):
            C = 8 * d
            qkv = torch.randn(64, L, 3 * C, device=dev).bfloat16().requires_grad_()
            kvc = torch.randn(64, 32, 2 * C, device=dev).bfloat16().requires_grad_()
            for mode in MODES:
                _lib.lib().mdm_dev_set_attn_bwd(mode)
                o = ops.attention(qkv, kvc, None, 8)
                go = torch.randn_like(o)
                for _ in range(3):
                    torch.autograd.grad(o, (qkv, kvc), go, retain_graph=True)
        torch.cuda.synchronize()
        sys.exit(0)
    cases = [
        (2, 256, 32, 8, 96, False),
        (3, 256, 32, 8, 64, True),
        (2, 200, 20, 4, 96, True),
        (2, 136, 0, 2, 96, False),
        (2, 64, 8, 2, 64, True),
        (40, 64, 8, 8, 64, True),
        (1, 1024, 32, 8, 64, True),
        (2, 1000, 20, 2, 64, True),
        (2, 512, 32, 2, 96, False),
    ]
    allok = True
    for c in cases:
        allok &= case(*c)
    print("ALL OK" if allok else "SOME FAILED")
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        timing(int(os.environ.get("KB_BATCH", "64")))

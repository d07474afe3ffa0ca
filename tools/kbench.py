#!/usr/bin/env python
"""Per-shape timing of the GEMM-class kernels on the UNet-64 (batch 64) shapes.  Development tool:
   gpurun -- python tools/kbench.py [fwd|wgrad|attn|all]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-mdm_amd"))
import torch  # noqa: E402

from mdm_hip import ops  # noqa: E402

B = int(os.environ.get("KB_BATCH", "64"))
DT = torch.bfloat16 if os.environ.get("KB_DTYPE", "bf16") == "bf16" else torch.float32
dev = torch.device("cuda:0")

# (name, H, Cin, Cout, ks, stride)
SHAPES = [
    ("3x3 256->256 @64", 64, 256, 256, 3, 1),
    ("3x3 512->512 @32", 32, 512, 512, 3, 1),
    ("3x3 768->768 @16", 16, 768, 768, 3, 1),
    ("3x3 512->256 @64", 64, 512, 256, 3, 1),
    ("3x3 1536->768 @16", 16, 1536, 768, 3, 1),
    ("3x3 1280->512 @32", 32, 1280, 512, 3, 1),
    ("3x3s2 256->256 @64", 64, 256, 256, 3, 2),
    ("1x1 768->3072 @16", 16, 768, 3072, 1, 1),
    ("1x1 3072->768 @16", 16, 3072, 768, 1, 1),
    ("1x1 768->2304 @16", 16, 768, 2304, 1, 1),
    ("1x1 768->768 @16", 16, 768, 768, 1, 1),
    ("1x1 512->2048 @32", 32, 512, 2048, 1, 1),
    ("1x1 512->1536 @32", 32, 512, 1536, 1, 1),
]


if os.environ.get("KB_SET") == "nested":   # the outer (256 x 256, 64 / 128 channel) levels of the nested 64+256 model; use KB_BATCH=16
    SHAPES = [
        ("3x3 64->64 @256", 256, 64, 64, 3, 1),
        ("3x3 128->64 @256", 256, 128, 64, 3, 1),
        ("3x3 64->128 @128", 128, 64, 128, 3, 1),
        ("3x3 128->128 @128", 128, 128, 128, 3, 1),
        ("3x3 256->128 @128", 128, 256, 128, 3, 1),
        ("1x1 128->64 @256", 256, 128, 64, 1, 1),
    ]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if os.environ.get("KB_TILE"):   # force the forward / input-gradient tile: 128128 / 256192 / 256256 (development knob 2)
        from mdm_hip import _lib
        _lib.lib().mdm_dev_set_knob(2, int(os.environ["KB_TILE"]))
    only = os.environ.get("KB_ONLY")
    for idx, (name, H, cin, cout, ks, stride) in enumerate(SHAPES):
        if only is not None and idx != int(only):
            continue
        x = torch.randn(B, H, H, cin, device=dev).to(DT)
        w = (torch.randn(cout, cin, ks, ks, device=dev) / (cin * ks * ks) ** 0.5)
        b = torch.randn(cout, device=dev)
        wf, wd, bp, cin_p, cout_p, kbf, kbd = ops.packed_weight(w, b, DT)
        Ho = (H - 1) // stride + 1
        y = torch.empty(B, Ho, Ho, cout, device=dev, dtype=DT)
        flops = 2.0 * B * Ho * Ho * cout * cin * ks * ks
        line = "%-22s" % name
        if what in ("fwd", "all"):
            act = int(os.environ.get("KB_ACT", "0"))          # 1: GELU epilogue writing y and the pre-activation (FFN up); 2: x gelu'(aux)
            ypre = torch.empty_like(y) if act == 1 else None
            aux = torch.randn_like(y) if act == 2 else None
            res = torch.randn_like(y) if os.environ.get("KB_RES") == "1" else None   # + residual (proj_out, conv2, FFN down)
            t = timeit(lambda: ops._conv_launch(x, wf, bp, res, aux, y, ypre, B, H, H, cin, Ho, Ho, cout, ks, stride, 0, act, kbf))
            line += "  fwd %7.3f ms %7.1f TF" % (t * 1e3, flops / t / 1e12)
        if what in ("dgrad", "all") and stride == 1:
            dx = torch.empty_like(x)
            t = timeit(lambda: ops._conv_launch(y, wd, None, None, None, dx, None, B, Ho, Ho, cout, H, H, cin, ks, 1, 0, 0, kbd))
            line += "  dgrad %7.3f ms %7.1f TF" % (t * 1e3, flops / t / 1e12)
        if what in ("wgrad", "all"):
            t = timeit(lambda: ops._wgrad_launch(x, y, B, H, H, cin, Ho, Ho, cout, ks, stride))
            line += "  wgrad %7.3f ms %7.1f TF" % (t * 1e3, flops / t / 1e12)
            if os.environ.get("KB_DBIAS") == "1":     # the same launch also producing the bias gradient (column sums of dY)
                db = torch.empty(cout, device=dev)
                t = timeit(lambda: ops._wgrad_launch(x, y, B, H, H, cin, Ho, Ho, cout, ks, stride, dbias=db))
                line += "  +dbias %7.3f ms" % (t * 1e3)
        print(line, flush=True)
    if what in ("rotate",):
        # the store-heavy 1x1 GEMMs with FRESH output buffers every launch (a ring of `KB_RING` buffers larger than the
        # 256 MB Infinity Cache), as in a train step -- rewriting one buffer lets the caches absorb the writes
        ring = int(os.environ.get("KB_RING", "6"))
        for idx, (name, H, cin, cout, ks, stride) in enumerate(SHAPES):
            if ks != 1:
                continue
            for act in (0, 1, 2, 3):   # 3 = no activation, + residual
                if act in (1, 2) and cout < cin * 2:
                    continue
                x = torch.randn(B, H, H, cin, device=dev).to(DT)
                w = (torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5)
                wf, wd, bp, cin_p, cout_p, kbf, kbd = ops.packed_weight(w, torch.randn(cout, device=dev), DT)
                ys = [torch.empty(B, H, H, cout, device=dev, dtype=DT) for _ in range(ring)]
                from mdm_hip import _lib
                byte = DT == torch.bfloat16 and _lib.lib().mdm_dev_ffn_aux_bytes() == 1   # gelu' as a byte code (round 6)
                yps = [torch.empty(B, H, H, cout * (2 if os.environ.get('KB_KNOB3_UNUSED') else 1), device=dev, dtype=torch.uint8 if byte else DT) for _ in range(ring)] if act == 1 else [None] * ring
                if act == 2 and byte:
                    auxs = [torch.randint(3, 250, (B, H, H, cout), device=dev, dtype=torch.uint8) for _ in range(ring)]
                else:
                    auxs = [torch.randn(B, H, H, cout, device=dev).to(DT) for _ in range(ring)] if act >= 2 else [None] * ring
                flops = 2.0 * B * H * H * cout * cin
                it = [0]
                def one():
                    j = it[0] % ring; it[0] += 1
                    if act == 3:
                        ops._conv_launch(x, wf, bp, auxs[j], None, ys[j], None, B, H, H, cin, H, H, cout, 1, 1, 0, 0, kbf)
                    else:
                        ops._conv_launch(x, wf, bp if act != 2 else None, None, auxs[j], ys[j], yps[j], B, H, H, cin, H, H, cout, 1, 1, 0, act, kbf)
                if os.environ.get("KB_KNOB3_UNUSED"):
                    _lib.lib().mdm_dev_set_knob(3, int(os.environ["KB_KNOB3_UNUSED"]))
                t = timeit(one, iters=3 * ring)
                it[0] = 0
                t1 = timeit(lambda: (it.__setitem__(0, 0), one())[1], iters=3 * ring)
                out_mb = (ys[0].numel() * 2 + (yps[0].numel() * yps[0].element_size() if act == 1 else 0)) / 1e6
                print("%-20s act=%d out %6.0f MB  fresh buffers %7.1f us %6.0f TF | same buffer %7.1f us %6.0f TF" % (
                    name, act, out_mb, t * 1e6, flops / t / 1e12, t1 * 1e6, flops / t1 / 1e12), flush=True)
    if what in ("stores",):
        # what the store phase of the 1x1 GEMMs costs: the same launch with the epilogue's global stores skipped (dev knob 1)
        from mdm_hip import _lib
        L = _lib.lib()
        for idx, (name, H, cin, cout, ks, stride) in enumerate(SHAPES):
            if ks != 1:
                continue
            for act in (0, 1):
                if act == 1 and cout < cin * 2:
                    continue
                x = torch.randn(B, H, H, cin, device=dev).to(DT)
                w = (torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5)
                wf, wd, bp, cin_p, cout_p, kbf, kbd = ops.packed_weight(w, torch.randn(cout, device=dev), DT)
                y = torch.empty(B, H, H, cout, device=dev, dtype=DT)
                ypre = torch.empty_like(y) if act == 1 else None
                flops = 2.0 * B * H * H * cout * cin
                line = "%-20s act=%d tile=%d " % (name, act, L.mdm_conv_fwd_tile(B * H * H, cout, 1))
                for tile in (0, 128128):
                    for knob1 in (0, 1):
                        L.mdm_dev_set_knob(2, tile); L.mdm_dev_set_knob(1, knob1)
                        t = timeit(lambda: ops._conv_launch(x, wf, bp, None, None, y, ypre, B, H, H, cin, H, H, cout, 1, 1, 0, act, kbf), iters=20)
                        line += " | %s %s %6.1f us %5.0f TF" % ("128x128" if tile else "model  ", "nostore" if knob1 else "store  ", t * 1e6, flops / t / 1e12)
                L.mdm_dev_set_knob(2, 0); L.mdm_dev_set_knob(1, 0)
                print(line, flush=True)
    if what in ("grouped",):
        G = int(os.environ.get("KB_GROUPS", "26"))
        for name, H, cin, cout, ks, stride in SHAPES:
            if ks != 1:
                continue
            M = B * H * H
            xs = [torch.randn(M, cin, device=dev).to(DT) for _ in range(G)]
            dys = [torch.randn(M, cout, device=dev).to(DT) for _ in range(G)]
            dws = [torch.zeros(cout, cin, device=dev) for _ in range(G)]
            dbs = [torch.zeros(cout, device=dev) for _ in range(G)]
            t = timeit(lambda: ops.wgrad_grouped(xs, dys, dws, dbs), iters=5)
            fl = 2.0 * G * M * cin * cout
            t1 = timeit(lambda: [ops._wgrad_launch(xs[i].view(B, H, H, cin), dys[i].view(B, H, H, cout), B, H, H, cin, H, H, cout, 1, 1, out=dws[i], dbias=dbs[i]) for i in range(G)], iters=3)
            print("%-22s grouped x%d %7.3f ms %7.1f TF   one-by-one (split + reduce) %7.3f ms %7.1f TF" % (name, G, t * 1e3, fl / t / 1e12, t1 * 1e3, fl / t1 / 1e12), flush=True)
    if what in ("adamw",):
        # the fused optimizer pass on a UNet-64-sized arena (415 M parameters): 5 reads + 5 writes of 4 bytes each
        n = 415_000_000
        p_, g_, m_, v_, e_ = [torch.zeros(n, device=dev) for _ in range(5)]
        g_.normal_()
        gn = torch.ones(1, device=dev)
        def step():
            ops.adamw_ema_step(p_, g_, m_, v_, e_, gn, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1, 2.0, 0.9999, zero_grad=True)
        for _ in range(2):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10e3
        print("adamw_ema %d M parameters: %.3f ms  %.0f GB/s" % (n // 1000000, t * 1e3, 40.0 * n / t / 1e9), flush=True)
        ws_ = torch.empty(1)
        t = timeit(lambda: ops.sumsq(g_))
        print("sumsq: %.3f ms  %.0f GB/s" % (t * 1e3, 4.0 * n / t / 1e9), flush=True)
    if what in ("pack",):
        # the optimizer tail's re-pack of every kernel-layout weight (ops.repack_all) on a UNet-64-like weight set
        shapes = [((256, 256, 3, 3), 7), ((512, 512, 3, 3), 6), ((768, 768, 3, 3), 10), ((256, 512, 3, 3), 3), ((512, 768, 3, 3), 1),
                  ((512, 1024, 3, 3), 2), ((512, 1280, 3, 3), 1), ((768, 1280, 3, 3), 1), ((768, 1536, 3, 3), 3),
                  ((2304, 768, 1, 1), 26), ((768, 768, 1, 1), 26), ((3072, 768, 1, 1), 26), ((768, 3072, 1, 1), 26),
                  ((1536, 512, 1, 1), 5), ((512, 512, 1, 1), 5), ((2048, 512, 1, 1), 5), ((512, 2048, 1, 1), 5)]
        ws = [torch.randn(*sh, device=dev) for sh, cnt in shapes for _ in range(cnt)]
        for w in ws:
            ops.packed_weight(w, None, DT)
        n = sum(w.numel() for w in ws)
        for _ in range(2):
            ops.repack_all(DT)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.repack_all(DT)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10e3
        print("pack %d weights, %.1f M parameters: %.3f ms  %.0f GB/s (4 B read + 2 x 2 B written per parameter)" % (len(ws), n / 1e6, t * 1e3, 8.0 * n / t / 1e9), flush=True)
        # spot check against the single-weight pack
        w = ws[0]
        a = [t_.clone() for t_ in ops.packed_weight(w, None, DT)[:2]]
        ops._wcache.clear()
        b = ops.packed_weight(w, None, DT)[:2]
        print("pack matches single-weight pack:", all(torch.equal(x_, y_) for x_, y_ in zip(a, b)), flush=True)
    if what in ("gn",):
        # GroupNorm(+SiLU) forward / backward at the shapes of the cc12m_64x64 U-Net: achieved GB/s over the minimal
        # traffic (fwd: read x + write y; bwd: read dy, x + write dx)
        for H, C in ((64, 256), (64, 512), (32, 512), (32, 768), (32, 1024), (32, 1280), (16, 768), (16, 1280), (16, 1536)):
            x = torch.randn(B, H, H, C, device=dev).to(DT).requires_grad_()
            gam = torch.randn(C, device=dev).requires_grad_()
            bet = torch.randn(C, device=dev).requires_grad_()
            t = timeit(lambda: ops.group_norm(x.detach(), gam.detach(), bet.detach(), 32, silu=True))
            y = ops.group_norm(x, gam, bet, 32, silu=True)
            gy = torch.randn_like(y)
            tb = timeit(lambda: torch.autograd.grad(y, (x, gam, bet), gy, retain_graph=True))
            nb = x.numel() * x.element_size()
            # the backward kernel(s) alone, launched directly (no autograd / Python between launches)
            from mdm_hip import _lib
            film = None
            stats = torch.empty((B, 32, 2), dtype=torch.float32, device=dev)
            coef = torch.empty((B, C, 2), dtype=torch.float32, device=dev)
            ws = ops._gn_ws(B, H * H, C, 32, dev)
            yy = torch.empty_like(x)
            xd = x.detach()
            _lib.check(_lib.lib().mdm_gn_fwd(ops._p(xd), ops._p(gam.detach()), ops._p(bet.detach()), None, ops._p(yy), ops._p(stats), ops._p(coef), ops._p(ws), B, H * H, C, 32, 1e-5, 1, ops._dt(xd), ops._stream()), "fwd")
            gmode = int(os.environ.get('KB_GN_MODE', '2'))   # 2: per-sample rows (deferred reduce), 1: atomics into a slot
            dxx = torch.empty_like(xd); dg = torch.zeros(B if gmode == 2 else 1, C, device=dev); db = torch.zeros_like(dg)
            dres = torch.randn_like(xd) if os.environ.get('KB_GN_RES') else None   # + the residual-branch gradient (as in the model)
            def direct():
                _lib.check(_lib.lib().mdm_gn_bwd(ops._p(gy), ops._p(xd), ops._p(gam.detach()), ops._p(bet.detach()), None, ops._p(stats), ops._p(coef), ops._p(dres), None, ops._p(dxx), ops._p(dg), ops._p(db), None, ops._p(ws), B, H * H, C, 32, 1, gmode, ops._dt(xd), ops._stream()), "bwd")
            def directf():
                _lib.check(_lib.lib().mdm_gn_fwd(ops._p(xd), ops._p(gam.detach()), ops._p(bet.detach()), None, ops._p(yy), ops._p(stats), ops._p(coef), ops._p(ws), B, H * H, C, 32, 1e-5, 1, ops._dt(xd), ops._stream()), "fwd")
            nrot = int(os.environ.get('KB_GN_COLD', '0'))   # > 0: rotate through this many sets of tensors (cold HBM, not the 256 MB MALL)
            if nrot:
                sets = [(torch.randn_like(xd), torch.randn_like(xd), torch.empty_like(xd), torch.randn_like(xd) if dres is not None else None, torch.empty_like(xd)) for _ in range(nrot)]
                L_ = _lib.lib()
                # KB_GN_FULL=1: as the ResNet's second norm runs it -- FiLM (+ its gradient) and a second residual-branch gradient
                full = os.environ.get('KB_GN_FULL') == '1'
                flm = (0.1 * torch.randn(B, 2 * C, device=dev)).to(DT) if full else None
                dflm = torch.empty_like(flm) if full else None
                dr2 = torch.randn_like(xd) if full else None
                def direct(i=[0]):
                    x_, gy_, dx_, dr_, y_ = sets[i[0] % nrot]; i[0] += 1
                    _lib.check(L_.mdm_gn_bwd(ops._p(gy_), ops._p(x_), ops._p(gam.detach()), ops._p(bet.detach()), ops._p(flm), ops._p(stats), ops._p(coef), ops._p(dr_), ops._p(dr2), ops._p(dx_), ops._p(dg), ops._p(db), ops._p(dflm), ops._p(ws), B, H * H, C, 32, 1, gmode, ops._dt(xd), ops._stream()), "bwd")
                def directf(i=[0]):
                    x_, gy_, dx_, dr_, y_ = sets[i[0] % nrot]; i[0] += 1
                    _lib.check(L_.mdm_gn_fwd(ops._p(x_), ops._p(gam.detach()), ops._p(bet.detach()), None, ops._p(y_), ops._p(stats), ops._p(coef), ops._p(ws), B, H * H, C, 32, 1e-5, 1, ops._dt(xd), ops._stream()), "fwd")
            for f in (direct, directf):
                for _ in range(3): f()
            e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): direct()
            e1.record()
            for _ in range(50): directf()
            e2.record()
            torch.cuda.synchronize()
            tb, t = e0.elapsed_time(e1) / 50e3, e1.elapsed_time(e2) / 50e3
            print("gn %dx%d C=%-5d (%6.1f MB)  fwd %7.3f ms %6.0f GB/s   bwd %7.3f ms %6.0f GB/s%s" % (H, H, C, nb / 1e6, t * 1e3, 2 * nb / t / 1e9, tb * 1e3, (4 if dres is not None else 3) * nb / tb / 1e9, "   (cold: %d buffer sets)" % nrot if nrot else ""), flush=True)
            if nrot:
                del sets
    if what in ("attn", "all"):
        for L, d in ((1024, 64), (256, 96)):
            C = 8 * d
            qkv = torch.randn(B, L, 3 * C, device=dev).to(DT).requires_grad_()
            kvc = torch.randn(B, 32, 2 * C, device=dev).to(DT).requires_grad_()
            fl = 4.0 * B * 8 * L * (L + 32) * d
            t = timeit(lambda: ops.attention(qkv.detach(), kvc.detach(), None, 8))
            o = ops.attention(qkv, kvc, None, 8)
            go = torch.randn_like(o)
            tb = timeit(lambda: torch.autograd.grad(o, (qkv, kvc), go, retain_graph=True))
            print("attn L=%d d=%d  fwd %7.3f ms %6.1f TF   bwd %7.3f ms %6.1f TF" % (L, d, t * 1e3, fl / t / 1e12, tb * 1e3, 2.5 * fl / tb / 1e12), flush=True)


if __name__ == "__main__":
    main()

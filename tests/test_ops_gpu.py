"""Per-kernel parity: every libmdm_hip entry point (called through the C ABI via mdm_hip.ops)
against the plain torch fp32 CPU op it replaces, forward and backward.

Tolerances (max-abs error relative to the largest reference magnitude):
  fp32 mode : 2e-5   (exact-fp32 MFMA; differences are summation order only)
  bf16 mode : 3e-2   (bf16 storage of activations/weights, fp32 accumulation)
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 3e-2}


def dev():
    return torch.device("cuda:0")


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def q(t, dtype):
    """round a CPU fp32 tensor through the compute dtype so both sides see the same inputs"""
    return t.to(dtype).float()


def nhwc(t_nchw, dtype):
    return t_nchw.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev())


def nchw(t_nhwc):
    return t_nhwc.float().cpu().permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize(
    "N,H,W,Cin,Cout,ks,stride",
    [
        (2, 16, 16, 32, 64, 3, 1),
        (3, 10, 12, 40, 136, 3, 1),     # ragged M / N / K tiles
        (2, 16, 16, 64, 64, 3, 2),      # downsample
        (2, 8, 8, 256, 768, 1, 1),      # qkv-style 1x1
        (1, 7, 5, 24, 8, 3, 1),         # small N config
        (2, 32, 32, 32, 32, 3, 1),
        (2, 12, 12, 136, 264, 3, 1),    # Cin not a multiple of the k-tile
        (8, 121, 119, 40, 136, 3, 1),   # 450 tiles of 256x256: the 8-wave big-tile kernel, ragged M / N / K
        (4, 128, 128, 72, 200, 1, 1),   # same, 1x1, exactly one wave of tiles
        (4, 256, 256, 16, 72, 3, 2),    # same, stride 2 (forward) + transposed gradient
        (3, 24, 20, 264, 328, 3, 1),    # Cout >= 256 and K >= 256: the 256x256 transpose-read wgrad tile, ragged
        (2, 16, 16, 512, 256, 1, 1),    # same, 1x1
        # Cin % 64 == 0 (bf16): the buffer-addressed k-loop (conv_gemm_bl_kernel), 128x128 and 256x256 tiles
        (3, 20, 24, 128, 136, 3, 1),    # 128x128, ragged M / N, image borders inside a tile
        (2, 18, 22, 64, 200, 3, 2),     # 128x128, stride 2
        (8, 127, 63, 128, 320, 3, 1),   # 256x256 forward (ragged M / N), 128x128 input gradient
        (8, 127, 63, 128, 264, 1, 1),   # 256x256, 1x1
        (8, 254, 126, 64, 264, 3, 2),   # 256x256, stride 2
        # stride 2 with Cout % 64 == 0, even H / W (bf16): the pixel-unshuffled input gradient (mdm_conv_s2_dgrad)
        (2, 32, 32, 256, 256, 3, 2),
        (4, 64, 48, 128, 192, 3, 2),    # ragged tiles of the 4 Cin = 512 wide output
        # power-of-two images / 1x1: the buffer-addressed wgrad (conv_wgrad_bl_kernel)
        (3, 4, 8, 64, 72, 3, 1),        # 128x128 tile, ragged reduction tail (M = 96), image borders everywhere
        (5, 9, 7, 64, 136, 1, 1),       # 128x128, 1x1, M = 315
        (2, 16, 16, 256, 320, 3, 1),    # 256x256 tile (K = 2304), ragged Cout
        (4, 128, 128, 64, 264, 1, 1),   # 256x256, 1x1 (M = 65536), K below one tile
        # 1024 tiles of 128x128 on 512 resident blocks: the persistent kernel's second output tile per block
        (8, 128, 128, 64, 128, 1, 1),   # single k-tile (K = 64)
        (8, 128, 128, 64, 128, 3, 1),
        # 32 / 64 channels on both sides, >= 65536 pixels, W % 64 == 0, H % 8 == 0 (bf16): the direct halo-tile kernel
        # (conv3x3_direct_kernel) forward and -- with the channel counts swapped -- input gradient
        (4, 128, 128, 32, 32, 3, 1),
        (2, 136, 256, 32, 64, 3, 1),    # 17 tile rows, two output-channel halves per block
        (4, 128, 128, 64, 32, 3, 1),    # 32-pixel-wide tiles, padded LDS rows
        (1, 256, 256, 64, 64, 3, 1),
        # 64 output channels, >= 262144 pixels in 8 x 32 tiles (bf16): the direct weight gradient (wgrad_direct_kernel):
        # one [64][taps * Cin] slab per block, both operands by LDS transpose reads of the staged pixel tile
        (4, 256, 256, 64, 64, 3, 1),    # 3x3: halo of x, image borders on all four sides of the tile grid
        (2, 264, 512, 64, 64, 3, 1),    # 33 tile rows: blocks with different tile counts
        (4, 256, 256, 128, 64, 1, 1),   # 1x1 shortcut of the skip concatenation: the split GEMM with 256 ranges
    ],
)
def test_conv_fwd_bwd(dtype, N, H, W, Cin, Cout, ks, stride):
    from mdm_hip import ops

    g = torch.Generator().manual_seed(0)
    x = q(torch.randn(N, Cin, H, W, generator=g), dtype).requires_grad_()
    w = q(torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks), dtype).requires_grad_()
    b = torch.randn(Cout, generator=g).requires_grad_()
    y_ref = F.conv2d(x, w, b, stride=stride, padding=(ks - 1) // 2)
    res = q(torch.randn(y_ref.shape, generator=g), dtype).requires_grad_()
    y_ref = y_ref + res
    gy = q(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(gy)

    xd = nhwc(x.detach(), dtype).requires_grad_()
    wd = w.detach().to(dev()).requires_grad_()
    bd = b.detach().to(dev()).requires_grad_()
    rd = nhwc(res.detach(), dtype).requires_grad_()
    y = ops.conv(xd, wd, bd, residual=rd, stride=stride)
    y.backward(nhwc(gy, dtype))
    tol = TOL[dtype]
    assert relerr(nchw(y), y_ref) < tol
    assert relerr(nchw(xd.grad), x.grad) < tol
    assert relerr(wd.grad, w.grad) < tol
    assert relerr(bd.grad, b.grad) < tol
    assert relerr(nchw(rd.grad), res.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize(
    "N,H,W,Cin,Cout",
    [
        (2, 32, 32, 256, 256),     # bf16: the sub-pixel input gradient adds the tap's gradient in its epilogue
        (4, 64, 48, 128, 192),     # same, ragged tiles
        (2, 16, 16, 64, 64),       # shapes that kernel does not take: the generic input gradient + one add
    ],
)
def test_conv_tap_joins_the_skip_gradient(dtype, N, H, W, Cin, Cout):
    """ops.conv_tap: y = conv(x) (3x3, stride 2) and a second reader of x (the skip connection of a down-sampling block,
    reference unet.py:566-567); dL/dx = conv input gradient + the second reader's gradient, from one kernel"""
    from mdm_hip import ops

    g = torch.Generator().manual_seed(3)
    x = q(torch.randn(N, Cin, H, W, generator=g), dtype).requires_grad_()
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9), dtype).requires_grad_()
    b = torch.randn(Cout, generator=g).requires_grad_()
    y_ref = F.conv2d(x, w, b, stride=2, padding=1)
    gy = q(torch.randn(y_ref.shape, generator=g), dtype)
    gs = q(torch.randn(x.shape, generator=g), dtype)
    torch.autograd.backward([y_ref, x * 1.0], [gy, gs])

    xd = nhwc(x.detach(), dtype).requires_grad_()
    wd = w.detach().to(dev()).requires_grad_()
    bd = b.detach().to(dev()).requires_grad_()
    h = xd * 1.0     # a non-leaf input, as in the model
    y, skip = ops.conv_tap(h, wd, bd, stride=2)
    assert skip.data_ptr() == h.data_ptr()
    torch.autograd.backward([y, skip], [nhwc(gy, dtype), nhwc(gs, dtype)])
    tol = TOL[dtype]
    assert relerr(nchw(y), y_ref) < tol
    assert relerr(nchw(xd.grad), x.grad) < tol
    assert relerr(wd.grad, w.grad) < tol
    assert relerr(bd.grad, b.grad) < tol
    # only the tap is read: the convolution's own output must still have received a gradient
    y2, skip2 = ops.conv_tap(nhwc(x.detach(), dtype).requires_grad_(), wd, bd, stride=2)
    with pytest.raises(Exception):
        skip2.sum().backward()


@pytest.mark.parametrize("N,H,W,Cin,ks", [(4, 256, 256, 64, 3), (2, 264, 512, 64, 3)])
def test_conv_wgrad_direct_kernel(N, H, W, Cin, ks):
    """The narrow weight gradients of the nested models' outer levels (64 output channels, a million pixels): the direct
    kernel is what runs, and its result equals the split GEMM's (development knob 8) to summation-order rounding -- both
    against the same bf16 operands; test_conv_fwd_bwd holds the same shapes against torch fp32."""
    from mdm_hip import _lib, ops

    g = torch.Generator().manual_seed(21)
    dtype = torch.bfloat16
    x = (torch.randn(N, H, W, Cin, generator=g)).to(dtype).to(dev())
    w = (torch.randn(64, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)).to(dev())
    b = torch.randn(64, generator=g).to(dev())
    gy = torch.randn(N, H, W, 64, generator=g).to(dtype).to(dev())

    def grads(knob):
        _lib.lib().mdm_dev_set_knob(8, knob)
        try:
            wd, bd = w.clone().requires_grad_(), b.clone().requires_grad_()
            y = ops.conv(x, wd, bd)
            ops.profile_begin()
            y.backward(gy)
            names = list(ops.profile_end(2516.6)["all_gemm_kernels"])
            return wd.grad, bd.grad, names
        finally:
            _lib.lib().mdm_dev_set_knob(8, 0)

    dw, db, names = grads(0)
    assert any("wgrad_direct_kernel" in n for n in names), names
    dw_ref, db_ref, names_ref = grads(1)
    assert not any("wgrad_direct_kernel" in n for n in names_ref), names_ref
    assert relerr(dw, dw_ref) < 2e-5 and relerr(db, db_ref) < 2e-5


@pytest.mark.parametrize("N,H,W,Cin,Cout,ks,stride", [(2, 16, 16, 256, 320, 3, 1), (3, 24, 20, 136, 200, 1, 1), (2, 18, 22, 64, 40, 3, 2),
                                                     (2, 9, 7, 32, 24, 3, 1)])
def test_conv_fp32_split_products(N, H, W, Cin, Cout, ks, stride):
    """MDM_F32_SPLIT (ops.fp32_split): fp32 tensors, every product as three bf16 MFMAs on bf16 hi + lo halves (common.hpp
    FragSplit) -- the sampling-at-reference-precision mode.  Against the torch fp32 convolution: an order of magnitude
    looser than the exact-fp32 MFMA path (2e-5), three orders tighter than bf16 (3e-2)."""
    from mdm_hip import _lib, ops

    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(N, Cout, (H - 1) // stride + 1, (W - 1) // stride + 1, generator=g)
    y_ref = F.conv2d(x, w, b, stride=stride, padding=(ks - 1) // 2) + res
    with torch.no_grad(), ops.fp32_split():
        y = ops.conv(nhwc(x, torch.float32), w.to(dev()), b.to(dev()), residual=nhwc(res, torch.float32), stride=stride)
        torch.cuda.synchronize()
        assert "bf16x3" in _lib.lib().mdm_last_gemm_kernel().decode()
        # the weight operand came in as hi / lo planes made ahead of time (MDM_F32_SPLIT_W); splitting it inside the k-loop
        # (MDM_F32_SPLIT) gives the same bits
        assert "weight planes" in _lib.lib().mdm_last_gemm_kernel().decode()
        ops._weight_planes_on = False
        try:
            y_loop = ops.conv(nhwc(x, torch.float32), w.to(dev()), b.to(dev()), residual=nhwc(res, torch.float32), stride=stride)
            torch.cuda.synchronize()
            name = _lib.lib().mdm_last_gemm_kernel().decode()
            assert "bf16x3" in name and "weight planes" not in name
        finally:
            ops._weight_planes_on = True
        assert torch.equal(y, y_loop)
    err = relerr(nchw(y), y_ref)
    assert err < 2e-4, err
    with torch.no_grad():   # and the switch is off again: the exact path
        y2 = ops.conv(nhwc(x, torch.float32), w.to(dev()), b.to(dev()), residual=nhwc(res, torch.float32), stride=stride)
        assert "bf16x3" not in _lib.lib().mdm_last_gemm_kernel().decode()
    assert relerr(nchw(y2), y_ref) < TOL[torch.float32]


def test_weight_planes_follow_a_repack():
    """the planes are cached on the packed weight and rebuilt when it is re-packed (a parameter update)"""
    from mdm_hip import ops

    g = torch.Generator().manual_seed(6)
    x = nhwc(torch.randn(2, 64, 8, 8, generator=g), torch.float32)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev())
    with torch.no_grad(), ops.fp32_split():
        y1 = ops.conv(x, w)
        y1b = ops.conv(x, w)
        w.mul_(2.0)
        y2 = ops.conv(x, w)
    assert torch.equal(y1, y1b)
    assert relerr(y2, 2 * y1) < 1e-6


def test_split_weight_planes_argument_checks():
    from mdm_hip import _lib

    L = _lib.lib()
    a = torch.zeros(64, device=dev())
    b = torch.zeros(64, device=dev())
    assert L.mdm_split_weight_planes(a.data_ptr(), b.data_ptr(), 60, None) < 0      # not whole runs of 8
    assert L.mdm_split_weight_planes(a.data_ptr(), a.data_ptr(), 64, None) < 0      # in place
    assert L.mdm_split_weight_planes(a.data_ptr(), b.data_ptr(), 64, None) == 0
    # MDM_F32_SPLIT_W with a reduction that is not a multiple of 8 (1x1, Cin = 4)
    x = torch.zeros(1, 2, 2, 4, device=dev()); y = torch.zeros(1, 2, 2, 8, device=dev()); w = torch.zeros(32, device=dev())
    assert L.mdm_conv_fwd(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), None, 1, 2, 2, 4, 2, 2, 8, 1, 1, 0, 0, 0, 3, None) < 0
    assert L.mdm_conv_fwd(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), None, 1, 2, 2, 4, 2, 2, 8, 1, 1, 0, 0, 0, 2, None) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,L,S,H,d,masked", [(2, 256, 32, 4, 96, True), (1, 1024, 32, 2, 64, False), (3, 64, 0, 2, 32, False)])
def test_attention_fp32_split_products(B, L, S, H, d, masked):
    """the attention forward in the same mode: Q K^T and P V as bf16x3 products, fp32 softmax"""
    from mdm_hip import ops

    g = torch.Generator().manual_seed(8)
    C = H * d
    qkv = torch.randn(B, L, 3 * C, generator=g) * 1.2
    kvc = torch.randn(B, S, 2 * C, generator=g) if S else None
    mask = (torch.rand(B, S, generator=g) > 0.3).float() if (S and masked) else None
    if mask is not None:
        mask[:, 0] = 1.0
    qh, kh, vh = [t.reshape(B, L, H, d).transpose(1, 2) for t in qkv.split(C, -1)]
    sc = 1.0 / math.sqrt(d)
    ref = torch.softmax(qh @ kh.transpose(-1, -2) * sc, -1) @ vh
    if S:
        kc, vc = [t.reshape(B, S, H, d).transpose(1, 2) for t in kvc.split(C, -1)]
        sx = qh @ kc.transpose(-1, -2) * sc
        if mask is not None:
            sx = sx.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
        ref = ref + torch.softmax(sx, -1) @ vc
    ref = ref.transpose(1, 2).reshape(B, L, C)
    with torch.no_grad(), ops.fp32_split():
        out = ops.attention(qkv.to(dev()), kvc.to(dev()) if S else None, mask.to(dev()) if mask is not None else None, H)
    err = relerr(out.float().cpu(), ref)
    assert err < 2e-4, err


@pytest.mark.parametrize("act,with_res", [(0, False), (0, True), (1, False), (2, False)])
@pytest.mark.parametrize(
    "N,H,W,Cin,Cout,ks",
    [
        (3, 16, 16, 512, 384, 1),      # 9 tiles on 9 blocks, the shortest reduction the kernel takes (8 k-tiles)
        (75, 32, 32, 576, 256, 1),     # 600 tiles: 2-3 per block, 9 k-tiles (the drain ends in the tile's last iteration)
        (10, 32, 16, 64, 128, 3),      # 3x3: image borders inside the tiles, one column tile
        (75, 32, 32, 64, 256, 3),      # 3x3, several tiles per block
    ],
)
def test_conv_gemm_epilogues(act, with_res, N, H, W, Cin, Cout, ks):
    """every epilogue variant of the forward / input-gradient GEMM launched directly through mdm_conv_fwd (bias, +residual,
    GELU + the saved operand of its backward, x gelu'(aux)), 1x1 and 3x3, several output tiles per block and fewer tiles
    than CUs, against the torch fp32 ops it replaces (unet.py:199-217,266-272)."""
    from mdm_hip import _lib, ops

    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    x = q(torch.randn(N, Cin, H, W, generator=g), dtype)
    w = q(torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks), dtype)
    b = torch.randn(Cout, generator=g)
    pre = q(F.conv2d(x, w, b, padding=(ks - 1) // 2), dtype)            # the conv output as the bf16 graph has it
    res = q(torch.randn(pre.shape, generator=g), dtype) if with_res else None
    aux = q(torch.randn(pre.shape, generator=g) * 1.5, dtype) if act == 2 else None

    def dgelu(t):
        a = t.clone().requires_grad_()
        return torch.autograd.grad(F.gelu(a).sum(), a)[0]

    # bf16 tensors: the operand the GELU launch leaves for its backward is ONE BYTE per element, the code
    # q = round(196 gelu'(pre)) + 28 (include/mdm_hip.h MDM_ACT_GELU; csrc/common.hpp DGeluCode)
    code = None
    if act == 1:
        y_ref = F.gelu(pre)
    elif act == 2:
        code = torch.round(dgelu(aux) * 196.0 + 28.0).clamp(0, 255).to(torch.uint8)
        y_ref = pre * dgelu(aux)
    else:
        y_ref = pre + (res if with_res else 0)
    xd = nhwc(x, dtype)
    wf, wd, bp, cin_p, cout_p, kbf, kbd = ops.packed_weight(w.to(dev()), b.to(dev()), dtype)
    y = torch.full((N, H, W, Cout), float("nan"), device=dev(), dtype=dtype)
    ypre = torch.full((N, H, W, Cout), 255, device=dev(), dtype=torch.uint8) if act == 1 else None
    auxd = code.permute(0, 2, 3, 1).contiguous().to(dev()) if act == 2 else None
    ops._conv_launch(xd, wf, bp, nhwc(res, dtype) if with_res else None, auxd, y, ypre,
                     N, H, W, Cin, H, W, Cout, ks, 1, 0, act, kbf)
    torch.cuda.synchronize()
    assert relerr(nchw(y), y_ref) < TOL[dtype]
    if act == 1:
        # decoded gelu' against the erf form of the bf16 pre-activation the kernel saw: half a code step (2.55e-3), the
        # polynomial's 5.1e-4, and the bf16 rounding of the pre-activation on the kernel's side (the reference `pre` here is
        # rounded from an fp32 convolution: the two roundings may differ by one ulp, |gelu''| <= 1.13 of it)
        got = (ypre.permute(0, 3, 1, 2).float().cpu() - 28.0) / 196.0
        err = (got - dgelu(pre)).abs()
        ulp = pre.abs().clamp(min=2.0 ** -10) * 2.0 ** -7
        assert bool((err <= 2.56e-3 + 5.2e-4 + 1.13 * ulp).all()), float(err.max())
        assert float(err.mean()) < 1.6e-3      # ~ a quarter step on average: the code is unbiased
    if act == 2:
        # ... and the decode side exactly: the product with the decoded byte, rounded once
        y_code = q(pre * ((code.float() - 28.0) / 196.0), dtype)
        assert relerr(nchw(y), y_code) < 5e-3     # (two bf16 roundings of slightly different fp32 sums)


def test_gelu_poly_against_erf_gelu():
    """The transcendental-free GELU / GELU' of the bf16 epilogues (csrc/common.hpp gelu_poly, dgelu_poly) against the erf
    forms (nn.GELU(), unet.py:270) on EVERY bf16 value in [-9, 9], evaluated by the kernels themselves: an identity 1x1
    convolution (exact in bf16: one nonzero product per output) feeds each value to the GELU epilogue.  Gates: gelu within
    1.5e-4 absolute + the bf16 rounding of the result; gelu' (decoded from its byte code) within 6e-4 + half a code step."""
    from mdm_hip import ops

    dtype = torch.bfloat16
    bits = torch.arange(0, 1 << 16, dtype=torch.int32)
    vals = bits.to(torch.int16).view(torch.bfloat16).float()
    vals = vals[torch.isfinite(vals) & (vals.abs() <= 9.0)]
    C = 64
    n = (vals.numel() + C - 1) // C
    n = (n + 255) // 256 * 256
    z = torch.zeros(n * C)
    z[:vals.numel()] = vals
    x = z.view(n, C).to(dtype).to(dev())
    w = torch.eye(C).view(C, C, 1, 1).to(dev())
    wf, wd, bp, cin_p, cout_p, kbf, kbd = ops.packed_weight(w, None, dtype)
    y = torch.empty((n, C), device=dev(), dtype=dtype)
    ypre = torch.empty((n, C), device=dev(), dtype=torch.uint8)
    ops._conv_launch(x, wf, None, None, None, y, ypre, n, 1, 1, C, 1, 1, C, 1, 1, 0, 1, kbf)
    torch.cuda.synchronize()
    zz = z.view(n, C).double()
    cdf = 0.5 * (1.0 + torch.erf(zz / math.sqrt(2.0)))
    gelu = zz * cdf
    dgelu = cdf + zz * torch.exp(-0.5 * zz * zz) / math.sqrt(2.0 * math.pi)
    e1 = (y.double().cpu() - gelu).abs()
    assert bool((e1 <= 1.5e-4 + gelu.abs() * 2.0 ** -8).all()), float((e1 - gelu.abs() * 2.0 ** -8).max())
    e2 = ((ypre.double().cpu() - 28.0) / 196.0 - dgelu).abs()
    assert bool((e2 <= 6e-4 + 0.5 / 196.0 + 1e-6).all()), float(e2.max())


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 256, 256), (3, 8, 16, 512, 256), (2, 16, 8, 768, 768)])
def test_upsample_conv_sub_pixel_form(N, H, W, Cin, Cout):
    """conv3x3(upsample2x(x)) computed from the low-resolution x (mdm_conv_up_fwd: per output phase a 2x2 correlation with
    the coinciding 3x3 taps summed), its input gradient (mdm_space_to_depth2x + mdm_conv_up_dgrad) and its weight / bias
    gradient (mdm_conv_wgrad_blocked: 16 of 36 (phase, tap) blocks, + mdm_upconv_wfold) against F.interpolate + F.conv2d
    (reference unet.py:567-569); bf16 (fp32 and unsupported shapes take upsample2x + conv, covered by test_streaming_helpers
    and test_conv_fwd_bwd)."""
    from mdm_hip import ops

    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(11)
    x = q(torch.randn(N, Cin, H, W, generator=g), dtype).requires_grad_()
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9), dtype).requires_grad_()
    b = torch.randn(Cout, generator=g).requires_grad_()
    y_ref = F.conv2d(F.interpolate(x, scale_factor=2), w, b, padding=1)
    gy = q(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(gy)
    xd = nhwc(x.detach(), dtype).requires_grad_()
    wd = w.detach().to(dev()).requires_grad_()
    bd = b.detach().to(dev()).requires_grad_()
    assert ops.upsample_conv_supported(xd, wd)
    y = ops.upsample_conv(xd, wd, bd)
    y.backward(nhwc(gy, dtype))
    tol = TOL[dtype]
    assert relerr(nchw(y), y_ref) < tol
    assert relerr(nchw(xd.grad), x.grad) < tol
    assert relerr(wd.grad, w.grad) < tol
    assert relerr(bd.grad, b.grad) < tol
    # ... and into a gradient arena (accumulate mode of the fold), on top of what is already there
    class Sink:
        def __init__(self, params):
            self.slots = {p.data_ptr(): torch.full_like(p, 0.5) for p in params}
        def slot(self, p):
            return self.slots.get(p.data_ptr())
        def ready(self, p):
            pass
    sink = Sink([wd, bd])
    ops.set_grad_sink(sink)
    try:
        xd2 = nhwc(x.detach(), dtype).requires_grad_()
        ops.upsample_conv(xd2, wd, bd).backward(nhwc(gy, dtype))
        ops.join_side_stream()
        torch.cuda.synchronize()
    finally:
        ops.set_grad_sink(None)
    assert relerr(sink.slots[wd.data_ptr()] - 0.5, w.grad) < tol
    assert relerr(sink.slots[bd.data_ptr()] - 0.5, b.grad) < tol
    assert relerr(nchw(xd2.grad), x.grad) < tol


@pytest.mark.parametrize("N,Cin,Cout,ks,silu,res", [(2, 768, 768, 1, False, True), (3, 768, 768, 3, True, True),
                                                    (2, 1536, 768, 3, True, False), (1, 256, 768, 1, False, False)])
def test_conv_with_group_norm_of_its_output(N, Cin, Cout, ks, silu, res):
    """mdm_conv_fwd_gn: y = conv(x) + bias (+ residual) and act(GroupNorm(y)) from one launch (16x16 images, 32 groups of 24
    channels), forward and every gradient -- y feeds both the norm and a second consumer, as the residual stream does
    (reference unet.py:310-311 proj_out -> ffn[0]) -- against F.conv2d + F.group_norm."""
    from mdm_hip import ops

    dtype = torch.bfloat16
    H = W = 16
    g = torch.Generator().manual_seed(5)
    x = q(torch.randn(N, Cin, H, W, generator=g), dtype).requires_grad_()
    w = q(torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks), dtype).requires_grad_()
    b = torch.randn(Cout, generator=g).requires_grad_()
    r = q(torch.randn(N, Cout, H, W, generator=g), dtype).requires_grad_() if res else None
    gamma = (1 + 0.3 * torch.randn(Cout, generator=g)).requires_grad_()
    beta = (0.3 * torch.randn(Cout, generator=g)).requires_grad_()
    y_ref = F.conv2d(x, w, b, padding=ks // 2)
    if res:
        y_ref = y_ref + r
    yq = q(y_ref, dtype)                       # the norm reads the stored (rounded) y on both sides
    yq = y_ref + (yq - y_ref).detach()
    n_ref = F.group_norm(yq, 32, gamma, beta, 1e-5)
    if silu:
        n_ref = F.silu(n_ref)
    gy, gn = q(torch.randn(y_ref.shape, generator=g), dtype), q(torch.randn(y_ref.shape, generator=g), dtype)
    torch.autograd.backward([y_ref, n_ref], [gy, gn])
    xd = nhwc(x.detach(), dtype).requires_grad_()
    wd = w.detach().to(dev()).requires_grad_()
    bd = b.detach().to(dev()).requires_grad_()
    rd = nhwc(r.detach(), dtype).requires_grad_() if res else None
    gd = gamma.detach().to(dev()).requires_grad_()
    btd = beta.detach().to(dev()).requires_grad_()
    assert ops.conv_gn_supported(xd, wd, gd, 32)
    y, yn = ops.conv_gn(xd, wd, bd, rd, gd, btd, 32, 1e-5, silu=silu)
    torch.autograd.backward([y, yn], [nhwc(gy, dtype), nhwc(gn, dtype)])
    tol = TOL[dtype]
    assert relerr(nchw(y), y_ref) < tol
    assert relerr(nchw(yn), n_ref) < tol
    assert relerr(nchw(xd.grad), x.grad) < tol
    assert relerr(wd.grad, w.grad) < tol
    assert relerr(bd.grad, b.grad) < tol
    assert relerr(gd.grad, gamma.grad) < tol
    assert relerr(btd.grad, beta.grad) < tol
    if res:
        assert relerr(nchw(rd.grad), r.grad) < tol
    # identical to the two-launch path (conv, then group_norm with the pass-through) up to the rounding of y
    xs = nhwc(x.detach(), dtype)
    y2 = ops.conv(xs, wd.detach(), bd.detach(), residual=rd.detach() if res else None)
    n2 = ops.group_norm(y2, gd.detach(), btd.detach(), 32, 1e-5, silu=silu)
    assert relerr(y2, y.detach()) < 8e-3       # one bf16 ulp: the unfused launch may pick another tile / accumulation order
    assert relerr(n2, yn.detach()) < 2e-2
    # shapes the fused epilogue does not take
    assert not ops.conv_gn_supported(nhwc(torch.randn(2, 768, 32, 32), dtype), wd, gd, 32)
    assert not ops.conv_gn_supported(nhwc(torch.randn(2, Cin, 16, 16), torch.float32), wd, gd, 32)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_padded_stem_and_head(dtype):
    """3-channel stem (input padded to a chunk) and 3-channel head (output padded)."""
    from mdm_hip import ops

    g = torch.Generator().manual_seed(1)
    x = q(torch.randn(2, 3, 16, 16, generator=g), dtype)
    w = q(torch.randn(32, 3, 3, 3, generator=g) / 5, dtype).requires_grad_()
    b = torch.randn(32, generator=g).requires_grad_()
    y_ref = F.conv2d(x, w, b, padding=1)
    gy = q(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(gy)
    wd, bd = w.detach().to(dev()).requires_grad_(), b.detach().to(dev()).requires_grad_()
    y = ops.conv(ops.to_nhwc(x.to(dev()), dtype), wd, bd)
    y.backward(nhwc(gy, dtype))
    tol = TOL[dtype]
    assert relerr(nchw(y), y_ref) < tol
    assert relerr(wd.grad, w.grad) < tol and relerr(bd.grad, b.grad) < tol

    h = q(torch.randn(2, 32, 16, 16, generator=g), dtype).requires_grad_()
    w2 = q(torch.randn(3, 32, 3, 3, generator=g) / 17, dtype).requires_grad_()
    b2 = torch.randn(3, generator=g).requires_grad_()
    o_ref = F.conv2d(h, w2, b2, padding=1)
    go = torch.randn(o_ref.shape, generator=g)
    o_ref.backward(go)
    hd = nhwc(h.detach(), dtype).requires_grad_()
    w2d, b2d = w2.detach().to(dev()).requires_grad_(), b2.detach().to(dev()).requires_grad_()
    o = ops.from_nhwc(ops.conv(hd, w2d, b2d), 3)
    o.backward(go.to(dev()))
    assert relerr(o.cpu(), o_ref) < tol
    assert relerr(nchw(hd.grad), h.grad) < tol
    assert relerr(w2d.grad, w2.grad) < tol and relerr(b2d.grad, b2.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,N", [
    (5, 128, 72),          # tiled GEMM (N not a multiple of 16)
    (64, 1024, 1536),      # time_layer shape (with MDM_HIP_SMALLM=1: the experimental split-K small-M kernel)
    (33, 256, 1024),       # ragged rows
    (7, 2048, 48),         # K > one round of the four waves
])
def test_linear_small_rows(dtype, M, K, N):
    from mdm_hip import ops

    g = torch.Generator().manual_seed(2)
    x = q(torch.randn(M, K, generator=g), dtype).requires_grad_()
    w = q(torch.randn(N, K, generator=g) / math.sqrt(K), dtype).requires_grad_()
    b = torch.randn(N, generator=g).requires_grad_()
    y_ref = F.linear(x, w, b)
    gy = q(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(gy)
    xd = x.detach().to(dtype).to(dev()).requires_grad_()
    wd, bd = w.detach().to(dev()).requires_grad_(), b.detach().to(dev()).requires_grad_()
    y = ops.linear(xd, wd, bd)
    y.backward(gy.to(dtype).to(dev()))
    tol = TOL[dtype]
    assert relerr(y.float().cpu(), y_ref) < tol
    assert relerr(xd.grad.float().cpu(), x.grad) < tol
    assert relerr(wd.grad, w.grad) < tol and relerr(bd.grad, b.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_ffn(dtype):
    from mdm_hip import ops

    g = torch.Generator().manual_seed(3)
    N, H, W, C = 2, 8, 8, 64
    x = q(torch.randn(N, C, H, W, generator=g), dtype).requires_grad_()
    r = q(torch.randn(N, C, H, W, generator=g), dtype).requires_grad_()
    w1 = q(torch.randn(4 * C, C, 1, 1, generator=g) / 8, dtype).requires_grad_()
    b1 = torch.randn(4 * C, generator=g).requires_grad_()
    w2 = q(torch.randn(C, 4 * C, 1, 1, generator=g) / 16, dtype).requires_grad_()
    b2 = torch.randn(C, generator=g).requires_grad_()
    y_ref = F.conv2d(F.gelu(F.conv2d(x, w1, b1)), w2, b2) + r
    gy = q(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(gy)
    ps = [t.detach().to(dev()).requires_grad_() for t in (w1, b1, w2, b2)]
    xd, rd = nhwc(x.detach(), dtype).requires_grad_(), nhwc(r.detach(), dtype).requires_grad_()
    y = ops.ffn(xd, *ps, residual=rd)
    y.backward(nhwc(gy, dtype))
    tol = TOL[dtype]
    assert relerr(nchw(y), y_ref) < tol
    assert relerr(nchw(xd.grad), x.grad) < tol
    for a, b_ in zip(ps, (w1, b1, w2, b2)):
        assert relerr(a.grad, b_.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,HW,C,G,film,silu", [
    (2, 64, 32, 32, False, True),      # 1 channel / group
    (3, 100, 192, 32, True, True),     # 6 channels / group: chunks straddle groups
    (2, 256, 768, 32, True, True),
    (2, 49, 1280, 32, False, False),
    (2, 1024, 64, 32, False, True),    # 2 channels / group
    # whole chunks per group and a small image: the single-kernel (register-resident) GroupNorm
    (2, 1024, 512, 32, True, True),    # 16 / group, 8 passes forward; backward 16 passes of a 512-thread block (bf16)
    (3, 256, 512, 32, False, False),   # no activation
    (2, 400, 1536, 32, True, True),    # 48 / group (one group per block), 4 passes
    (2, 256, 1536, 32, True, True),    # 48 / group at the 16x16 level's size
    (2, 256, 256, 32, False, True),    # 8 / group, 8 groups per block
    # large images: partial -> apply with the per-block finalize of a channel slice
    (2, 4096, 256, 32, True, True),    # the 64x64 level: 64-channel slices, 8 groups each
    (2, 4096, 768, 32, False, True),   # skip-concat width: 24 / group -> 96-channel slices
    (3, 1024, 1280, 32, True, True),   # 40 / group -> 80-channel slices (backward only: forward is register-resident)
    (2, 2304, 32, 32, False, True),    # nested outer level: 1 channel / group, the slice is the whole row
    (5, 900, 128, 32, False, False),   # 4 / group; pixel count not a multiple of anything
])
def test_group_norm(dtype, N, HW, C, G, film, silu):
    from mdm_hip import ops

    g = torch.Generator().manual_seed(4)
    H = int(math.isqrt(HW)); W = HW // H
    assert H * W == HW
    x = q(torch.randn(N, C, H, W, generator=g) * 1.7 + 0.6, dtype).requires_grad_()
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_()
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_()
    fl = q(0.5 * torch.randn(N, 2 * C, generator=g), dtype).requires_grad_() if film else None
    y_ref = F.group_norm(x, G, gamma, beta, 1e-5)
    if film:
        y_ref = y_ref * (1 + fl[:, :C, None, None]) + fl[:, C:, None, None]
    if silu:
        y_ref = F.silu(y_ref)
    gy = q(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(gy)
    xd = nhwc(x.detach(), dtype).requires_grad_()
    gd, bd = gamma.detach().to(dev()).requires_grad_(), beta.detach().to(dev()).requires_grad_()
    fd = fl.detach().to(dtype).to(dev()).requires_grad_() if film else None
    y = ops.group_norm(xd, gd, bd, G, 1e-5, film=fd, silu=silu)
    y.backward(nhwc(gy, dtype))
    tol = TOL[dtype]
    assert relerr(nchw(y), y_ref) < tol
    assert relerr(nchw(xd.grad), x.grad) < tol
    assert relerr(gd.grad, gamma.grad) < tol and relerr(bd.grad, beta.grad) < tol
    if film:
        assert relerr(fd.grad.float().cpu(), fl.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,HW,C,G", [(2, 256, 768, 32), (2, 1024, 64, 32), (2, 1024, 512, 32)])
def test_group_norm_residual_passthrough(dtype, N, HW, C, G):
    """h = x + f(norm(x)): the residual branch is routed through the norm's second output, so its gradient is added
    inside the GroupNorm backward kernel (single-kernel and three-kernel paths)."""
    from mdm_hip import ops

    g = torch.Generator().manual_seed(5)
    H = int(math.isqrt(HW)); W = HW // H
    x = q(torch.randn(N, C, H, W, generator=g), dtype).requires_grad_()
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_()
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_()
    w_res = q(torch.randn(N, C, H, W, generator=g), dtype)          # makes the two branch gradients different
    y_ref = F.silu(F.group_norm(x, G, gamma, beta, 1e-5)) * 1.5 + x * w_res
    gy = q(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(gy)
    xd = nhwc(x.detach(), dtype).requires_grad_()
    gd, bd = gamma.detach().to(dev()).requires_grad_(), beta.detach().to(dev()).requires_grad_()
    y, xr = ops.group_norm(xd, gd, bd, G, 1e-5, silu=True, passthrough=True)
    out = y.float() * 1.5 + xr.float() * nhwc(w_res, dtype).float()
    out.backward(nhwc(gy, dtype).float())
    tol = TOL[dtype]
    assert relerr(nchw(out), y_ref) < tol
    assert relerr(nchw(xd.grad), x.grad) < tol
    assert relerr(gd.grad, gamma.grad) < tol and relerr(bd.grad, beta.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,HW,C,G", [(2, 256, 768, 32), (2, 1024, 512, 32), (2, 4096, 256, 32)])
def test_group_norm_two_passthroughs(dtype, N, HW, C, G):
    """x is also a skip activation: a third output carries it to a consumer outside the block, and BOTH extra gradients
    (residual branch, skip connection) are added inside the GroupNorm backward kernel (mdm_gn_bwd dres / dres2;
    single-kernel and two-kernel paths) -- what the autograd engine did with one accumulation kernel per skip."""
    from mdm_hip import ops

    g = torch.Generator().manual_seed(6)
    H = int(math.isqrt(HW)); W = HW // H
    x = q(torch.randn(N, C, H, W, generator=g), dtype).requires_grad_()
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_()
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_()
    w_res, w_skip = q(torch.randn(N, C, H, W, generator=g), dtype), q(torch.randn(N, C, H, W, generator=g), dtype)
    y_ref = F.silu(F.group_norm(x, G, gamma, beta, 1e-5)) * 1.5 + x * w_res + (x * w_skip).tanh()
    gy = q(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(gy)
    xd = nhwc(x.detach(), dtype).requires_grad_()
    gd, bd = gamma.detach().to(dev()).requires_grad_(), beta.detach().to(dev()).requires_grad_()
    y, xr, xs = ops.group_norm(xd, gd, bd, G, 1e-5, silu=True, passthrough=2)
    out = y.float() * 1.5 + xr.float() * nhwc(w_res, dtype).float() + (xs.float() * nhwc(w_skip, dtype).float()).tanh()
    out.backward(nhwc(gy, dtype).float())
    tol = TOL[dtype]
    assert relerr(nchw(out), y_ref) < tol
    assert relerr(nchw(xd.grad), x.grad) < tol
    assert relerr(gd.grad, gamma.grad) < tol and relerr(bd.grad, beta.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_layer_norm(dtype):
    from mdm_hip import ops

    g = torch.Generator().manual_seed(5)
    x = q(torch.randn(3, 9, 2048, generator=g) * 2 + 0.3, dtype).requires_grad_()
    gamma = (1 + 0.3 * torch.randn(2048, generator=g)).requires_grad_()
    beta = (0.2 * torch.randn(2048, generator=g)).requires_grad_()
    y_ref = F.layer_norm(x, (2048,), gamma, beta, 1e-5)
    gy = q(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(gy)
    xd = x.detach().to(dtype).to(dev()).requires_grad_()
    gd, bd = gamma.detach().to(dev()).requires_grad_(), beta.detach().to(dev()).requires_grad_()
    y = ops.layer_norm(xd, gd, bd, 1e-5)
    y.backward(gy.to(dtype).to(dev()))
    tol = TOL[dtype]
    assert relerr(y.float().cpu(), y_ref) < tol
    assert relerr(xd.grad.float().cpu(), x.grad) < tol
    assert relerr(gd.grad, gamma.grad) < tol and relerr(bd.grad, beta.grad) < tol


def _attn_ref(qkv, kvc, mask, heads):
    """qkv [B, L, 3C], kvc [B, S, 2C]: the reference formulation (models/unet.py:276-307)."""
    import unet_oracle as O

    B, L, C3 = qkv.shape
    C = C3 // 3
    t = qkv.transpose(1, 2)
    out = O.attention_core(t[:, :C], t[:, C:2 * C], t[:, 2 * C:], heads)
    if kvc is not None:
        kt = kvc.transpose(1, 2)
        out = out + O.attention_core(t[:, :C], kt[:, :C], kt[:, C:], heads, mask)
    return out.transpose(1, 2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,L,S,H,d,masked", [
    (2, 64, 8, 8, 32, False),
    (2, 256, 32, 8, 96, False),       # UNet-64 level 2 geometry
    (1, 1024, 32, 8, 64, True),       # UNet-64 level 1 geometry, masked text
    (2, 200, 77, 2, 64, True),        # ragged L and S (two key tiles for the text)
    (2, 128, 0, 4, 32, False),        # no cross attention
    # short sequences take the one-block-per-head backward (attn_bwd_small_kernel: L <= 256, S <= 64, bf16)
    (3, 256, 32, 8, 64, True),        # masked text, 4 query / key tiles
    (2, 200, 20, 4, 96, True),        # ragged L (last tile partial), ragged S
    (2, 72, 64, 2, 128, False),       # d = 128, a full text tile, L barely over one tile
    (2, 136, 0, 2, 96, False),        # no cross attention, second 128-key / 128-query pass partial
    (40, 64, 8, 8, 32, True),         # B * H = 320 blocks: more heads than CUs (second round of blocks)
    # csrc/attn32.hpp (d = 64 / 96, at most 32 text keys): a wave owns 32 keys / queries
    (2, 256, 32, 8, 96, True),        # UNet-64 level 2 geometry, masked text
    (40, 96, 20, 8, 64, True),        # 320 heads, three 32-row tiles, ragged text
    (2, 1000, 20, 2, 64, True),       # streaming kernels: ragged L (last stage of 64 partial, last 256-row block partial)
    (2, 320, 32, 2, 64, False),       # streaming kernels, second block of 256 rows mostly empty
    (2, 33, 3, 1, 64, True),          # one row into the second tile
])
def test_attention(dtype, B, L, S, H, d, masked, request):
    from mdm_hip import _lib, ops

    # the library picks the one-block-per-head backward only when B * H fills the chip: force it for the small cases
    _lib.lib().mdm_dev_set_attn_bwd(2)
    request.addfinalizer(lambda: _lib.lib().mdm_dev_set_attn_bwd(0))

    g = torch.Generator().manual_seed(6)
    C = H * d
    qkv = q(torch.randn(B, L, 3 * C, generator=g) * 1.2, dtype).requires_grad_()
    kvc = q(torch.randn(B, S, 2 * C, generator=g) * 1.2, dtype).requires_grad_() if S else None
    mask = None
    if masked:
        mask = torch.ones(B, S)
        mask[0, S // 2:] = 0
        mask[-1, 1:3] = 0
    # spike one key against one query so the running max jumps mid-stream (online-softmax rescale path)
    with torch.no_grad():
        qkv[0, L // 2, :d] *= 6.0
        qkv[0, (3 * L) // 4, C:C + d] = qkv[0, L // 2, :d] * 0.5
    o_ref = _attn_ref(qkv, kvc, mask, H)
    go = q(torch.randn(o_ref.shape, generator=g), dtype)
    o_ref.backward(go)
    qd = qkv.detach().to(dtype).to(dev()).requires_grad_()
    kd = kvc.detach().to(dtype).to(dev()).requires_grad_() if S else None
    md = mask.to(dev()) if mask is not None else None
    o = ops.attention(qd, kd, md, H)
    o.backward(go.to(dtype).to(dev()))
    tol = TOL[dtype]
    assert relerr(o.float().cpu(), o_ref) < tol
    assert relerr(qd.grad.float().cpu(), qkv.grad) < tol
    if S:
        assert relerr(kd.grad.float().cpu(), kvc.grad) < tol
    if dtype == torch.bfloat16:
        # every other backward path on the same case (mdm_hip_dev.h: 1 = the two streaming kernels on 16x16x32
        # MFMAs, 3 = one block per head on 16x16x32, 4 = the streaming kernels on 32x32x16 of csrc/attn32.hpp where the
        # shape allows): all must agree with the reference
        for mode in (1, 3, 4):
            _lib.lib().mdm_dev_set_attn_bwd(mode)
            qd2 = qkv.detach().to(dtype).to(dev()).requires_grad_()
            kd2 = kvc.detach().to(dtype).to(dev()).requires_grad_() if S else None
            ops.attention(qd2, kd2, md, H).backward(go.to(dtype).to(dev()))
            assert relerr(qd2.grad.float().cpu(), qkv.grad) < tol, mode
            if S:
                assert relerr(kd2.grad.float().cpu(), kvc.grad) < tol, mode


@pytest.mark.parametrize("dtype", DTYPES)
def test_streaming_helpers(dtype):
    from mdm_hip import ops

    g = torch.Generator().manual_seed(7)
    tol = TOL[dtype]
    a = q(torch.randn(2, 24, 6, 10, generator=g), dtype).requires_grad_()
    b = q(torch.randn(2, 40, 6, 10, generator=g), dtype).requires_grad_()
    ref = F.silu(torch.cat([F.interpolate(a, scale_factor=2), F.interpolate(b, scale_factor=2)], 1))
    gy = q(torch.randn(ref.shape, generator=g), dtype)
    ref.backward(gy)
    ad, bd = nhwc(a.detach(), dtype).requires_grad_(), nhwc(b.detach(), dtype).requires_grad_()
    out = ops.silu(ops.concat(ops.upsample2x(ad), ops.upsample2x(bd)))
    out.backward(nhwc(gy, dtype))
    assert relerr(nchw(out), ref) < tol
    assert relerr(nchw(ad.grad), a.grad) < tol and relerr(nchw(bd.grad), b.grad) < tol

    x = q(torch.randn(3, 7, 64, generator=g), dtype).requires_grad_()
    m = torch.ones(3, 7); m[1, 4:] = 0
    ref = (m.unsqueeze(-1) * x).sum(1) / m.sum(1, keepdim=True)
    gy = q(torch.randn(ref.shape, generator=g), dtype)
    ref.backward(gy)
    xd = x.detach().to(dtype).to(dev()).requires_grad_()
    y = ops.masked_mean(xd, m.to(dev()))
    y.backward(gy.to(dtype).to(dev()))
    assert relerr(y.float().cpu(), ref) < tol and relerr(xd.grad.float().cpu(), x.grad) < tol

    t = torch.tensor([0.0, 3.0, 999.0])
    fr = torch.exp(torch.arange(16, dtype=torch.float) * -(math.log(10000) / 16))
    e = ops.sincos_embedding(t.to(dev()), fr.to(dev()), dtype)
    ang = t[:, None] * fr[None]
    assert relerr(e.float().cpu(), torch.cat([ang.sin(), ang.cos()], 1)) < max(tol, 2e-4)
    s = ops.add(ad.detach(), ad.detach())
    assert relerr(nchw(s), 2 * a.detach()) < tol
    c = ops.cast(xd.detach(), torch.float32)
    assert c.dtype == torch.float32 and relerr(c.cpu(), x.detach()) < tol


@pytest.mark.parametrize("G,M,Cin,Cout,bias", [
    (4, 4096, 256, 512, True),       # 256x256 tiles
    (3, 5000, 136, 264, False),      # ragged M / Cin / Cout, no bias
    (26, 4096, 768, 768, True),      # the attention stack's proj_out geometry (short reduction for test time)
    (5, 4160, 64, 128, True),        # 128x128 tiles
    (1, 4096, 768, 2304, True),      # a single problem is legal too
])
def test_grouped_weight_gradient(G, M, Cin, Cout, bias):
    """mdm_conv_wgrad_grouped: the weight / bias gradients of G same-shape 1x1 convolutions in ONE launch, ADDED into
    their destinations (no split of the pixel reduction, no slabs)"""
    from mdm_hip import ops

    g = torch.Generator().manual_seed(12)
    xs = [q(torch.randn(M, Cin, generator=g), torch.bfloat16) for _ in range(G)]
    dys = [q(torch.randn(M, Cout, generator=g), torch.bfloat16) for _ in range(G)]
    init = [torch.randn(Cout, Cin, generator=g) for _ in range(G)]
    binit = [torch.randn(Cout, generator=g) for _ in range(G)]
    dws = [t.clone().to(dev()) for t in init]
    dbs = [t.clone().to(dev()) for t in binit] if bias else None
    ops.wgrad_grouped([t.to(torch.bfloat16).to(dev()) for t in xs], [t.to(torch.bfloat16).to(dev()) for t in dys], dws, dbs)
    for i in range(G):
        ref = dys[i].double().t() @ xs[i].double()
        assert relerr(dws[i] - init[i].to(dev()), ref) < 2e-5, i
        if bias:
            assert relerr(dbs[i] - binit[i].to(dev()), dys[i].double().sum(0)) < 2e-5, i


@pytest.mark.parametrize("L,C,mid_flush", [(24, 512, False),   # 24 x 16 tiles of 128x128: goes out as ONE grouped launch
                                          (8, 256, True)])    # two small queues: flushed as individual launches
def test_deferred_grouped_wgrad_equals_immediate(L, C, mid_flush):
    """a chain of same-shape 1x1 convolutions with the gradient sink: queued + grouped weight gradients (flushed at a
    wgrad_flush_point and at the end) == the immediate per-layer launches"""
    from mdm_hip import ops

    class Sink:
        def __init__(self, params):
            self.slots = {p.data_ptr(): torch.zeros_like(p) for p in params}
            self.seen = []

        def slot(self, p):
            return self.slots.get(p.data_ptr())

        def ready(self, p):
            self.seen.append(p.data_ptr())

    g = torch.Generator().manual_seed(2)
    N, H = 8, 32
    ws = [(torch.randn(C, C, 1, 1, generator=g) / (2 * C ** 0.5)).to(dev()).requires_grad_() for _ in range(L)]
    bs = [torch.zeros(C, device=dev()).requires_grad_() for _ in range(L)]
    x0 = torch.randn(N, H, H, C, generator=g).to(dev()).to(torch.bfloat16)
    out = []
    for deferred in (False, True):
        sink = Sink(ws + bs)
        ops.set_grad_sink(sink)
        ops.enable_async_wgrad(True)
        ops.enable_deferred_wgrad(deferred)
        x = x0.clone().requires_grad_()
        h = ops.wgrad_flush_point(x)
        for i in range(L):
            h = ops.conv(h, ws[i], bs[i], residual=h)
            if mid_flush and i == 3:
                h = ops.wgrad_flush_point(h)
        h.float().square().mean().backward()
        ops.flush_wgrad_queue()
        ops.join_side_stream()
        torch.cuda.synchronize()
        assert sorted(sink.seen) == sorted(p.data_ptr() for p in ws + bs)
        out.append(([sink.slots[w.data_ptr()].clone() for w in ws], [sink.slots[b.data_ptr()].clone() for b in bs], x.grad.clone()))
    ops.set_grad_sink(None)
    ops.enable_async_wgrad(False)
    ops.enable_deferred_wgrad(False)
    for a, b in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
        assert relerr(b, a) < 2e-5          # split vs unsplit reduction order
    assert torch.equal(out[0][2], out[1][2])


@pytest.mark.parametrize("B,S,D,couts", [(4, 32, 128, [512, 512, 256, 512]), (64, 32, 2048, [1536] * 3 + [1024] * 2)])
def test_text_kv_of_all_layers_at_once(B, S, D, couts):
    """ops.text_kv (one multi-LayerNorm + grouped GEMMs for every attention layer's key / value projection of the text
    states) == the layers one by one with torch (LayerNorm + Linear), outputs and every gradient; bf16"""
    from mdm_hip import ops

    g = torch.Generator().manual_seed(9)
    dt = torch.bfloat16
    cond = q(torch.randn(B, S, D, generator=g) * 1.5 + 0.2, dt).requires_grad_()
    layers, gys = [], []
    for c in couts:
        layers.append([(1 + 0.3 * torch.randn(D, generator=g)).requires_grad_(), (0.2 * torch.randn(D, generator=g)).requires_grad_(),
                       (torch.randn(c, D, generator=g) / D ** 0.5).requires_grad_(), (0.1 * torch.randn(c, generator=g)).requires_grad_()])
        gys.append(q(torch.randn(B, S, c, generator=g), dt))
    refs = [F.linear(q(F.layer_norm(cond, (D,), lw, lb, 1e-5), dt), q(w, dt), b) for lw, lb, w, b in layers]
    sum((r * gy).sum() for r, gy in zip(refs, gys)).backward()
    cd = cond.detach().to(dt).to(dev()).requires_grad_()
    dl = [[t.detach().to(dev()).requires_grad_() for t in quad] for quad in layers]
    assert ops.text_kv_supported(cd, dl)
    outs = ops.text_kv(cd, dl)
    sum((o.float() * gy.to(dev())).sum() for o, gy in zip(outs, gys)).backward()
    tol = TOL[dt]
    for o, r in zip(outs, refs):
        assert relerr(o.float().cpu(), r) < tol
    assert relerr(cd.grad.float().cpu(), cond.grad) < tol
    for quad_d, quad in zip(dl, layers):
        for a, b in zip(quad_d, quad):
            assert relerr(a.grad, b.grad) < tol


@pytest.mark.parametrize("R,D,couts", [(64, 1024, [512] * 5 + [1024] * 5 + [1536] * 7), (3, 128, [64, 128, 64])])
def test_linears_sharing_one_input(R, D, couts):
    """ops.shared_input_linears (every ResNet's time_layer on silu(temb) as grouped GEMMs) == the layers one by one"""
    from mdm_hip import ops

    g = torch.Generator().manual_seed(10)
    dt = torch.bfloat16
    x = q(torch.randn(R, D, generator=g), dt).requires_grad_()
    layers = [[(torch.randn(c, D, generator=g) / D ** 0.5).requires_grad_(), (0.1 * torch.randn(c, generator=g)).requires_grad_()] for c in couts]
    gys = [q(torch.randn(R, c, generator=g), dt) for c in couts]
    refs = [F.linear(x, q(w, dt), b) for w, b in layers]
    sum((r * gy).sum() for r, gy in zip(refs, gys)).backward()
    xd = x.detach().to(dt).to(dev()).requires_grad_()
    dl = [[t.detach().to(dev()).requires_grad_() for t in pair] for pair in layers]
    assert ops.shared_input_linears_supported(xd, dl)
    outs = ops.shared_input_linears(xd, dl)
    sum((o.float() * gy.to(dev())).sum() for o, gy in zip(outs, gys)).backward()
    tol = TOL[dt]
    for o, r in zip(outs, refs):
        assert relerr(o.float().cpu(), r) < tol
    assert relerr(xd.grad.float().cpu(), x.grad) < tol
    for pd, pr in zip(dl, layers):
        assert relerr(pd[0].grad, pr[0].grad) < tol and relerr(pd[1].grad, pr[1].grad) < tol


def test_fails_loudly_without_gpu_tensor():
    from mdm_hip import _lib, ops

    with pytest.raises(_lib.MdmHipError):
        ops.silu(torch.randn(4, 8))


@pytest.mark.parametrize("deferred", [False, True])
def test_group_norm_backward_in_sample_chunks(deferred, request):
    """two-kernel GroupNorm backward walking the batch in chunks of samples (mdm_dev_set_gn_chunk_mb: the second kernel's
    reads of x and dy then come out of the Infinity Cache) == the whole batch at once; atomic and per-sample-row
    parameter gradients, FiLM gradients, residual-branch gradient"""
    from mdm_hip import _lib, ops

    class Sink:
        def __init__(self, params):
            self.slots = {p.data_ptr(): torch.zeros_like(p) for p in params}

        def slot(self, p):
            return self.slots.get(p.data_ptr())

        def ready(self, p):
            pass

    request.addfinalizer(lambda: (_lib.lib().mdm_dev_set_gn_chunk_mb(160), ops.set_grad_sink(None), ops.enable_deferred_wgrad(False)))
    g = torch.Generator().manual_seed(9)
    N, H, C = 16, 8, 192          # 6 channels per group: not register-resident -> partial + apply kernels
    gam = (1 + 0.1 * torch.randn(C, generator=g)).to(dev()).requires_grad_()
    bet = (0.1 * torch.randn(C, generator=g)).to(dev()).requires_grad_()
    x0 = torch.randn(N, H, H, C, generator=g).to(dev()).to(torch.bfloat16)
    fl = (0.2 * torch.randn(N, 2 * C, generator=g)).to(dev()).to(torch.bfloat16)
    res = []
    for mb in (-1, 0):            # never chunk / chunk whenever the batch allows it (4 chunks of 4 samples)
        _lib.lib().mdm_dev_set_gn_chunk_mb(mb)
        sink = Sink([gam, bet]) if deferred else None
        ops.set_grad_sink(sink)
        ops.enable_deferred_wgrad(deferred)
        gam.grad = bet.grad = None
        x = x0.clone().requires_grad_()
        f = fl.clone().requires_grad_()
        h = x + ops.group_norm(x, gam, bet, 32, film=f, silu=True)     # the residual gradient joins inside the backward
        (h.float() * torch.linspace(-1, 1, C, device=dev())).square().mean().backward()
        ops.flush_wgrad_queue()
        torch.cuda.synchronize()
        pg = [sink.slots[p.data_ptr()].clone() for p in (gam, bet)] if deferred else [gam.grad.clone(), bet.grad.clone()]
        res.append((x.grad.float().clone(), f.grad.float().clone(), pg))
    assert relerr(res[1][0], res[0][0]) < 2e-3 and relerr(res[1][1], res[0][1]) < 2e-3
    for a_, b_ in zip(res[0][2], res[1][2]):
        assert float(a_.abs().max()) > 0 and relerr(b_, a_) < 1e-4


@pytest.mark.parametrize("H,C,film", [(16, 768, False), (16, 1536, True), (32, 512, True), (64, 256, False)])
def test_deferred_group_norm_param_grads_equal_atomic_path(H, C, film):
    """with the gradient sink + deferral the GroupNorm backward stores per-sample rows and ONE multi-layer reduce adds
    them into the slots (flush point / end of backward) == the immediate atomic accumulation; two steps, so the pooled
    rows and the cached descriptor table are reused"""
    from mdm_hip import ops

    class Sink:
        def __init__(self, params):
            self.slots = {p.data_ptr(): torch.full_like(p, 0.5) for p in params}   # non-zero: the reduce ADDS
            self.seen = []

        def slot(self, p):
            return self.slots.get(p.data_ptr())

        def ready(self, p):
            self.seen.append(p.data_ptr())

    g = torch.Generator().manual_seed(5)
    N, L = 6, 3
    gam = [(1 + 0.1 * torch.randn(C, generator=g)).to(dev()).requires_grad_() for _ in range(L)]
    bet = [(0.1 * torch.randn(C, generator=g)).to(dev()).requires_grad_() for _ in range(L)]
    x0 = torch.randn(N, H, H, C, generator=g).to(dev()).to(torch.bfloat16)
    fl = (0.2 * torch.randn(N, 2 * C, generator=g)).to(dev()).to(torch.bfloat16) if film else None
    out = []
    for deferred in (False, True):
        sink = Sink(gam + bet)
        ops.set_grad_sink(sink)
        ops.enable_deferred_wgrad(deferred)
        for step in range(2):
            x = x0.clone().requires_grad_()
            f = fl.clone().requires_grad_() if film else None
            h = x
            for i in range(L):
                h = ops.group_norm(h, gam[i], bet[i], 32, film=f, silu=(i % 2 == 0))
                if i == 0:
                    h = ops.wgrad_flush_point(h)
            h.float().square().mean().backward()
            ops.flush_wgrad_queue()
        torch.cuda.synchronize()
        assert sorted(sink.seen) == sorted([p.data_ptr() for p in gam + bet] * 2)
        out.append(([sink.slots[p.data_ptr()].clone() for p in gam + bet], x.grad.clone(), f.grad.clone() if film else None))
    ops.set_grad_sink(None)
    ops.enable_deferred_wgrad(False)
    for a, b in zip(out[0][0], out[1][0]):
        assert relerr(b, a) < 1e-5          # atomic order vs fixed order
    assert torch.equal(out[0][1], out[1][1])
    if film:
        assert torch.equal(out[0][2], out[1][2])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_repack_all_matches_single_weight_pack(dtype):
    """the optimizer tail's one-launch re-pack (mdm_pack_weights_multi: 32x32xtaps bricks, 16-byte loads and stores)
    == the per-weight pack, for 1x1 and 3x3 weights incl. the block-major 3x3 reduction order"""
    from mdm_hip import ops
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 32, 1, 1), (96, 160, 1, 1), (128, 64, 3, 3), (64, 192, 3, 3), (256, 256, 3, 3), (32, 32, 3, 3)]
    ws = [torch.randn(*s, generator=g).to(dev()) for s in shapes]
    ops._wcache.clear()
    for w in ws:
        ops.packed_weight(w, None, dtype)
    for w in ws:                       # an optimizer step: new values, same storage
        w.mul_(0.5).add_(1.0)
    ops.repack_all(dtype)
    multi = [[t.clone() if t is not None else None for t in ops.packed_weight(w, None, dtype)[:2]] for w in ws]
    ops._wcache.clear()
    for w, m in zip(ws, multi):
        single = ops.packed_weight(w, None, dtype)[:2]
        for a, b in zip(m, single):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a, b), tuple(w.shape)


@pytest.mark.parametrize("case", ["conv3x3_res", "conv1x1_res", "ffn"])
def test_small_problem_split_k_matches_unsplit(case):
    """sampling at batch 1: too few output tiles to fill the chip, so the reduction is split over blocks (partial fp32
    tiles + one sum/epilogue launch).  Same result as the unsplit launch up to the order of the fp32 sum -- forward and
    every gradient (the FFN backward exercises the x gelu'(aux) epilogue of the second stage) -- and the split path is
    the one that ran."""
    from mdm_hip import ops
    g = torch.Generator().manual_seed(17)
    N, H = 1, 16
    if case == "conv3x3_res":
        C = 768
        w = (torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(dev()).requires_grad_()
    elif case == "conv1x1_res":
        C = 768
        w = (torch.randn(C, 4 * C, 1, 1, generator=g) / (2 * C ** 0.5)).to(dev()).requires_grad_()
    else:
        # wide -> narrow -> wide, so that the LONG reductions are the ones with the GELU epilogues: the up-projection
        # (GELU + pre-activation store) forward and the x gelu'(aux) input gradient of the down-projection backward
        C = 3072
        w = (torch.randn(768, C, 1, 1, generator=g) / C ** 0.5).to(dev()).requires_grad_()
        w2 = (torch.randn(C, 768, 1, 1, generator=g) / (2 * 768 ** 0.5)).to(dev()).requires_grad_()
        b2 = (0.1 * torch.randn(C, generator=g)).to(dev()).requires_grad_()
    b = (0.1 * torch.randn(w.shape[0], generator=g)).to(dev()).requires_grad_()
    x0 = torch.randn(N, H, H, w.shape[1], generator=g).to(dev()).to(torch.bfloat16)
    r0 = torch.randn(N, H, H, C, generator=g).to(dev()).to(torch.bfloat16)

    def run():
        x, r = x0.clone().requires_grad_(), r0.clone().requires_grad_()
        if case == "ffn":
            y = ops.ffn(x, w, b, w2, b2, r)
            ps = (x, r, w, b, w2, b2)
        else:
            y = ops.conv(x, w, b, residual=r)
            ps = (x, r, w, b)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(dev()).to(y.dtype)
        return [y.detach().float()] + [t.float() for t in torch.autograd.grad(y, ps, gy)]

    real_plan = ops._conv_plan
    seen = []

    def spy(M, Cout, K, dt):
        sp = real_plan(M, Cout, K, dt)
        seen.append(sp[0])
        return sp

    ops._conv_plan = spy
    try:
        split = run()
        ops._conv_plan = lambda M, Cout, K, dt: (1, 0)
        plain = run()
    finally:
        ops._conv_plan = real_plan
    assert sum(1 for v in seen if v > 1) >= (2 if case == "ffn" else 1), seen   # the launches this case is about were split
    for a, b_ in zip(split, plain):
        assert relerr(a, b_) < 8e-3     # bf16 outputs: an fp32 sum in another order moves a few results by one ulp (2^-8)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dropout_counter_based_mask(dtype):
    """ops.dropout (nn.Dropout of a ResNet block, models/unet.py:208,234): kept elements are scaled by 1 / (1 - p), the
    rest are zero, the keep rate matches p, the mask is a pure function of (seed, counter) -- same seed, same mask; the
    next call draws a fresh one -- and backward applies the mask of ITS forward call"""
    from mdm_hip import ops

    p = 0.3
    x = (torch.randn(4, 16, 16, 64) + 3.0).to(dtype).to(dev()).requires_grad_()
    ops.seed_dropout(1234)
    y1 = ops.dropout(x, p, True)
    y2 = ops.dropout(x, p, True)
    kept = y1 != 0
    rate = float(kept.float().mean())
    assert abs(rate - (1 - p)) < 0.02, rate
    assert torch.allclose(y1[kept].float(), (x.detach()[kept].float() / (1 - p)).to(dtype).float(), rtol=1e-2 if dtype == torch.bfloat16 else 1e-6)
    assert not torch.equal(y1 != 0, y2 != 0)                    # the stream advanced
    ops.seed_dropout(1234)
    assert torch.equal(ops.dropout(x, p, True), y1)              # replayable
    gy = torch.ones_like(y2)
    (gx,) = torch.autograd.grad(y2, x, gy)
    assert torch.equal(gx != 0, y2 != 0) and abs(float(gx[gx != 0].float().mean()) - 1 / (1 - p)) < 1e-2
    assert ops.dropout(x, p, False) is x and ops.dropout(x, 0.0, True) is x


def test_resnet_block_with_dropout_trains():
    """UNet config with dropout > 0 (reference ResNetConfig.dropout): forward + backward run in training mode, eval mode
    is deterministic and equals the dropout-free model"""
    import parity_cases as PC
    from mdm_hip import ops

    model, _, _ = PC.build_module("mini_unet")
    model = model.to(dev())
    for m in model.modules():
        if hasattr(m, "config") and hasattr(m.config, "dropout"):
            m.config.dropout = 0.25
    inp = PC.inputs("mini_unet")
    args = (inp["x"].to(dev()), inp["times"].to(dev()), inp["cond"].to(dev()), inp["mask"].to(dev()))
    model.eval()
    with torch.no_grad():
        e1, e2 = model(*args), model(*args)
    assert torch.equal(e1, e2)
    model.train()
    ops.seed_dropout(7)
    t1 = model(*args)
    t2 = model(*args)
    assert not torch.equal(t1, t2) and bool(torch.isfinite(t1).all())
    t1.square().mean().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())

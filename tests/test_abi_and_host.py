"""CPU: the C-ABI library loads and exports every symbol of include/mdm_hip.h; host-side logic
(configs, module construction, state_dict layout, fail-loud behaviour) works without a GPU."""
import ctypes
import os

import pytest
import torch

import parity_cases as PC


def test_library_exports_every_declared_symbol():
    from mdm_hip import _lib

    protos = _lib.header_prototypes()
    names = [p[0] for p in protos]
    assert len(names) >= 25 and len(set(names)) == len(names)
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), n
    L = _lib.lib()
    assert L.mdm_abi_version() == _lib.ABI_VERSION == 5


def test_plan_functions_are_host_only():
    from mdm_hip import _lib

    L = _lib.lib()
    splits, ws = ctypes.c_int(0), ctypes.c_size_t(0)
    assert L.mdm_conv_wgrad_plan(64 * 64 * 64, 256, 2304, 1, ctypes.byref(splits), ctypes.byref(ws)) == 0
    # weight slabs + a 64-float header + bias-gradient partials (up to 4 rows per split)
    assert splits.value >= 1 and ws.value == (splits.value * 256 * 2304 + 64 + max(4 * splits.value, 64) * 256) * 4
    assert L.mdm_gn_plan(4, 256, 768, 32, ctypes.byref(ws)) == 0 and ws.value > 0
    # invalid arguments are reported, not executed
    assert L.mdm_conv_wgrad_plan(1, 1, 1, 1, None, None) < 0
    assert b"splits_out" in L.mdm_last_error()


def test_tile_and_split_cost_models():
    """host-side launch planning (no GPU): GEMM tile choice and wgrad tile / split count"""
    from mdm_hip import _lib

    L = _lib.lib()
    BF16, F32 = 1, 0
    # forward / dgrad tile = BM * 1000 + BN
    assert L.mdm_conv_fwd_tile(16384, 768, BF16) == 256192      # N = 768 at M = 16384: exactly one round of 256x192 tiles
    assert L.mdm_conv_fwd_tile(16384, 2304, BF16) == 256192
    assert L.mdm_conv_fwd_tile(262144, 256, BF16) == 256256     # 1024 tiles = 4 whole rounds
    assert L.mdm_conv_fwd_tile(65536, 512, BF16) == 256256
    assert L.mdm_conv_fwd_tile(2048, 1536, BF16) == 128128      # too few rows for a round of the large tiles
    assert L.mdm_conv_fwd_tile(4096, 8, BF16) == 128032 and L.mdm_conv_fwd_tile(4096, 64, BF16) == 128064
    assert L.mdm_conv_fwd_tile(16384, 768, F32) == 128128       # fp32 (parity mode): one tile shape
    # wgrad: blocks = tiles x splits must not spill a nearly empty extra round over the resident blocks
    for (M, Cout, K) in [(16384, 768, 3072), (16384, 3072, 768), (65536, 512, 4608), (262144, 256, 2304), (65536, 2048, 512)]:
        sp, ws = ctypes.c_int(0), ctypes.c_size_t(0)
        assert L.mdm_conv_wgrad_plan(M, Cout, K, BF16, ctypes.byref(sp), ctypes.byref(ws)) == 0
        te = L.mdm_conv_wgrad_tile(M, Cout, K, BF16)
        assert te in (128, 256) and 1 <= sp.value <= 64
        tiles = -(-Cout // te) * -(-K // te)
        slots = 256 if te == 256 else 512
        blocks = tiles * sp.value
        rounds = -(-blocks // slots)
        assert blocks >= 0.8 * rounds * slots, (M, Cout, K, te, sp.value, blocks)
        mt = -(-M // 64)
        per = -(-mt // sp.value)
        assert -(-mt // per) == sp.value                        # canonical: no empty split
    assert L.mdm_conv_wgrad_tile(4096, 64, 64, BF16) == 128     # 256 needs both output dims >= 192
    assert L.mdm_conv_wgrad_tile(16384, 768, 3072, F32) == 128


def test_forward_split_k_plan():
    """host-side planning of the small-problem split (mdm_conv_fwd_plan; 256 CUs assumed without a GPU): training shapes
    never split; sampling shapes at batch 1-4 do, with >= 6 k-tiles per range, <= 16 ranges, and only when the cut
    saves >= 16 k-tiles of serial walk"""
    from mdm_hip import _lib

    L = _lib.lib()
    BF16, F32 = 1, 0

    def plan(M, Cout, K, dt=BF16):
        sp, ws = ctypes.c_int(0), ctypes.c_size_t(0)
        assert L.mdm_conv_fwd_plan(M, Cout, K, dt, ctypes.byref(sp), ctypes.byref(ws)) == 0
        assert ws.value == (sp.value * M * Cout * 4 if sp.value > 1 else 0)
        return sp.value

    for (M, Cout, K) in [(16384, 768, 6912), (65536, 512, 4608), (262144, 256, 2304), (16384, 3072, 768), (16384, 768, 3072)]:
        assert plan(M, Cout, K) == 1, (M, Cout, K)            # batch 64: the output tiles fill the chip
    for (M, Cout, K) in [(1024, 768, 6912), (1024, 768, 13824), (4096, 512, 4608), (1024, 768, 3072), (256, 768, 6912)]:
        sp = plan(M, Cout, K)
        nt = K // 64
        assert 2 <= sp <= 16 and nt // sp >= 6 and nt - -(-nt // sp) >= 16, (M, Cout, K, sp)
    assert plan(1024, 768, 768) == 1                            # 12 k-tiles: a second launch costs more than it saves
    assert plan(1024, 3072, 768) == 1                           # 192 tiles already give most CUs a block
    assert plan(1024, 768, 6912, F32) == 1 and plan(1024, 64, 6912) == 1 and plan(1024, 768, 6900) == 1


def test_invalid_arguments_rejected_before_launch():
    from mdm_hip import _lib

    L = _lib.lib()
    rc = L.mdm_conv_fwd(None, None, None, None, None, None, None, 1, 8, 8, 8, 8, 8, 8, 3, 1, 0, 0, 0, 1, None)
    assert rc < 0
    rc = L.mdm_attn_fwd(None, None, None, None, None, None, None, 1, 64, 0, 8, 32, 1, None)
    assert rc < 0


def test_product_path_refuses_cpu_tensors():
    """no CPU fallback: the module constructs on CPU (parameters) but cannot run there"""
    from mdm_hip import _lib

    model, _, _ = PC.build_module("mini_unet")
    inp = PC.inputs("mini_unet")
    with pytest.raises(_lib.MdmHipError):
        model(inp["x"], inp["times"], inp["cond"], inp["mask"])


def test_config_string_parsing_matches_reference_conventions():
    from mdm_hip import UNetConfig

    c = UNetConfig(resolution_channels="64,128,256", num_resnets_per_resolution="2", attention_levels="1,2",
                   num_attention_layers="0,1,5")
    assert c.resolution_channels == [64, 128, 256]
    assert c.num_resnets_per_resolution == [2, 2, 2]
    assert c.attention_levels == [1, 2] and c.num_attention_layers == [0, 1, 5]
    d = UNetConfig()
    assert d.resolution_channels == [128, 256, 256, 512, 1024] and d.num_attention_layers == [1] * 5


def test_shipped_architectures_have_the_reference_parameter_counts():
    """SURVEY.md / BASELINE.md: 461.4 M (UNet-64), 476.6 M (nested-256), 481.0 M (nested-1024)"""
    import mdm_hip
    from mdm_hip import configs

    with torch.device("meta"):
        n64 = sum(p.numel() for p in mdm_hip.UNet(3, 3, configs.unet64_config()).parameters())
        n256 = sum(p.numel() for p in mdm_hip.NestedUNet(3, 3, configs.nested256_config()).parameters())
        m1024 = mdm_hip.NestedUNet(3, 3, configs.nested1024_config())
        n1024 = sum(p.numel() for p in m1024.parameters())
    assert round(n64 / 1e6, 1) == 461.4
    assert round(n256 / 1e6, 1) == 476.6
    assert round(n1024 / 1e6, 1) == 481.0
    assert m1024.nest_ratio == [16, 4]


def test_state_dict_layout_of_unet64():
    import mdm_hip
    from mdm_hip import configs

    with torch.device("meta"):
        m = mdm_hip.UNet(3, 3, configs.unet64_config())
    sd = m.state_dict()
    assert len(sd) == 713
    assert sd["down_blocks.1.attn.0.qkv.weight"].shape == (1536, 512, 1, 1)
    assert sd["down_blocks.2.attn.3.kv_cond.weight"].shape == (1536, 2048)
    assert sd["up_blocks.0.resnets.0.conv1.weight"].shape == (768, 1536, 3, 3)
    assert sd["mid_blocks.0.attn.0.ffn.3.weight"].shape == (768, 3072, 1, 1)
    assert "t_emb" not in sd and sd["cond_layers.scale.1.weight"].shape == (1024, 1024)


def test_module_is_deepcopyable_and_checkpoint_roundtrips(tmp_path):
    import copy

    model, _, sd = PC.build_module("mini_nested")
    clone = copy.deepcopy(model)  # ModelEma does this (reference models/model_ema.py:16)
    assert all(torch.equal(v, sd[k]) for k, v in clone.state_dict().items())
    f = str(tmp_path / "ckpt.pth")
    model.save(f, other_items={"batch_num": 7})
    fresh, _, _ = PC.build_module("mini_nested", seed=5)
    rest = fresh.load(f)
    assert rest["batch_num"] == 7
    assert all(torch.equal(v, sd[k]) for k, v in fresh.state_dict().items())


def test_philox_reference_known_answers():
    """oracle/philox_ref.py (the host replay of the device noise generator) against the published Philox4x32-10
    known-answer vectors of the Random123 distribution (kat_vectors: counter / key all zero, all ones, pi digits)"""
    import philox_ref as P

    def run(c, k):
        return [int(x[0]) for x in P.philox4x32_10([c[0]], [c[1]], [c[2]], [c[3]], k[0], k[1])]

    assert run((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = 0xFFFFFFFF
    assert run((f, f, f, f), (f, f)) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert run((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    z = P.normals(1 << 18, seed=42)
    assert abs(float(z.mean())) < 1e-2 and abs(float(z.std()) - 1.0) < 1e-2


def test_no_mfma_result_is_read_back_early_on_a_branch_edge():
    """tools/mfma_hazard_scan.py over the ISA of the two MFMA-bearing sources: no read of an MFMA's destination registers
    within 6 wait states on ANY path, taken branch edges included.  ROCm 7.2's hazard recognizer missed exactly that in
    conv3x3_direct_kernel<64, 64, 8, 32, 4> (stale accumulators, caught by the GPU parity test; the kernel now fences its
    k-loop from its epilogue) -- this keeps a recompile from reintroducing it silently.  The same pass checks that no kernel
    of the two sources spills more than 24 vector registers (today's worst: 18), that the bias-gradient dot products of the
    weight-gradient kernel stay behind scalar branches, and -- round 6 -- that no GEMM epilogue has a vector load between the
    wide stores of its chunked store loop (on gfx950's single in-order vmcnt such a load makes every store wait for the
    acknowledgement of the one before it: 2.1 ms per train step in rounds 3-5; the round-5 ISA trips this check 760 times).
    Cross-compiles, no GPU needed."""
    import importlib.util
    import shutil

    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    spec = importlib.util.spec_from_file_location(
        "mfma_hazard_scan", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "mfma_hazard_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = __import__("sys").argv
    __import__("sys").argv = ["mfma_hazard_scan.py"]
    try:
        assert mod.main() == 0
    finally:
        __import__("sys").argv = old

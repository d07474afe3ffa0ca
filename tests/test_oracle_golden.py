"""CPU: pin the oracle (oracle/unet_oracle.py) to the reference.

(1) against the committed golden vectors that oracle/make_golden.py produced by running the
    REAL reference (apple/ml-mdm) in the build container;
(2) when /root/reference is present, against the reference executed live.
"""
import os

import pytest
import torch

import parity_cases as PC
import unet_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", PC.ALL_CASES)
def test_oracle_matches_golden(name):
    gold = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    _, _, sd = PC.build_module(name)
    # the regenerated parameters are the ones the reference was run with
    for k, s in gold["param_sum"].items():
        assert abs(float(sd[k].double().sum()) - s) <= 1e-9 * max(1.0, abs(s)), k
    outs, grads = PC.oracle_run(name)
    for o, r in zip(outs, gold["outputs"]):
        assert O.rel_l2(o, r) < 1e-6
    for k, n in gold["grad_norm"].items():
        g = grads[k].double()
        assert abs(float(g.norm()) - n) <= 1e-4 * max(n, 1e-6), k
        probe = float((g * PC.probe_for(k, g.shape)).sum())
        assert abs(probe - gold["grad_probe"][k]) <= 2e-4 * max(n, 1e-6) * (g.numel() ** 0.5), k
    for k, full in gold["grad_full"].items():
        assert O.rel_l2(grads[k], full) < 1e-4, k


@pytest.mark.parametrize("name", PC.CASES)
def test_oracle_fp64_noise_floor(name):
    """fp32 oracle vs fp64 oracle: the noise floor every fp32 implementation lives above"""
    o32, _ = PC.oracle_run(name, torch.float32, with_grad=False)
    o64, _ = PC.oracle_run(name, torch.float64, with_grad=False)
    for a, b in zip(o32, o64):
        assert O.rel_l2(a, b) < 1e-5


@pytest.mark.parametrize("name", list(PC.FULL))
def test_oracle_matches_full_size_golden(name):
    """the SHIPPED architectures at full size (461 M / 477 M / 481 M parameters): oracle == what the real reference
    produced for tests/golden/full_size.pt (outputs through their summaries; every parameter gradient through its norm
    and a seeded random projection).  nested1024 covers the x / std input normalisation of its 256 level."""
    gold = torch.load(os.path.join(GOLD, "full_size.pt"), weights_only=False)[name]
    _, sd = PC.full_module(name)
    for k, s in gold["param_sum"].items():
        assert abs(float(sd[k].double().sum()) - s) <= 1e-9 * max(1.0, abs(s)), k
    inp = PC.full_inputs(name)
    with_grad = "grad_norm" in gold
    leaf = {k: v.clone().requires_grad_(with_grad) for k, v in sd.items()}
    with torch.set_grad_enabled(with_grad):
        outs = PC.as_list(O.model_forward(leaf, PC.full_cfg(name), inp["x"], inp["times"], inp["cond"], inp["mask"]))
    for i, (o, g) in enumerate(zip(outs, gold["outputs"])):
        PC.check_summary(o, g, "%s.out%d" % (name, i), 2e-6)
    if with_grad:
        PC.loss_of(outs, inp["gys"]).backward()
        norms = sorted(gold["grad_norm"].values())
        floor = 1e-2 * norms[len(norms) // 2]
        for k, n in gold["grad_norm"].items():
            g = leaf[k].grad.double()
            assert abs(float(g.norm()) - n) <= 1e-4 * max(n, floor), k
            probe = float((g * PC.probe_for(k, g.shape)).sum())
            assert abs(probe - gold["grad_probe"][k]) <= 4e-4 * max(n, floor), k


@pytest.mark.reference
@pytest.mark.parametrize("name", ["mini_unet", "mini_nested", "mini_nested2"] + PC.EXTRA_CASES)
def test_oracle_matches_live_reference(name):
    import dataclasses

    import ref_import

    R = ref_import.load()

    def conv(cfg):
        d = {f.name: getattr(cfg, f.name) for f in dataclasses.fields(cfg)}
        rc = R.unet.ResNetConfig(**dataclasses.asdict(d.pop("resnet_config")))
        inner = d.pop("inner_config", None)
        if inner is None:
            return R.unet.UNetConfig(resnet_config=rc, **d)
        icfg = conv(inner)
        cls = R.nested_unet.Nested2UNetConfig if hasattr(icfg, "inner_config") else R.nested_unet.NestedUNetConfig
        return cls(resnet_config=rc, inner_config=icfg, **d)

    _, cfg, sd = PC.build_module(name)
    rcfg = conv(cfg)
    ref = (R.nested_unet.NestedUNet if hasattr(rcfg, "inner_config") else R.unet.UNet)(3, 3, rcfg)
    ref.load_state_dict(sd, strict=True)
    inp = PC.inputs(name)
    with torch.no_grad():
        yr = PC.as_list(ref(inp["x"], inp["times"], inp["cond"], inp["mask"], inp["micros"]))
    yo, _ = PC.oracle_run(name, with_grad=False)
    for a, b in zip(yo, yr):
        assert O.rel_l2(a, b) < 1e-6

"""Generate tests/golden/*.pt from the REAL reference (apple/ml-mdm) -- build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

For every case of tests/parity_cases.py: load the case's parameters into the reference's own
``UNet`` / ``NestedUNet`` (through ``state_dict`` -- which also proves key/shape compatibility),
run the reference forward + backward on CPU fp32 and store
  outputs          full tensors
  grad_norm[k]     ||dL/dp_k||            for every parameter
  grad_probe[k]    <dL/dp_k, probe_k>     (probe regenerated from the key at test time)
  grad_full[k]     full gradient for a few small tensors
  param_sum[k]     float64 sum of every parameter (regeneration check)
TEST INFRASTRUCTURE: not imported by the product.
"""
import dataclasses
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("ml-mdm_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
sys.dont_write_bytecode = True

import parity_cases as PC  # noqa: E402
import ref_import  # noqa: E402


def to_ref_cfg(R, cfg):
    d = {f.name: getattr(cfg, f.name) for f in dataclasses.fields(cfg)}
    rc = R.unet.ResNetConfig(**dataclasses.asdict(d.pop("resnet_config")))
    inner = d.pop("inner_config", None)
    if inner is None:
        return R.unet.UNetConfig(resnet_config=rc, **d)
    icfg = to_ref_cfg(R, inner)
    cls = R.nested_unet.Nested2UNetConfig if hasattr(icfg, "inner_config") else R.nested_unet.NestedUNetConfig
    return cls(resnet_config=rc, inner_config=icfg, **d)


def main():
    R = ref_import.load()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name in PC.CASES:
        _, cfg, sd = PC.build_module(name)
        rcfg = to_ref_cfg(R, cfg)
        cls = R.nested_unet.NestedUNet if hasattr(rcfg, "inner_config") else R.unet.UNet
        ref = cls(3, 3, rcfg)
        missing, unexpected = ref.load_state_dict(sd, strict=True)
        assert not missing and not unexpected
        inp = PC.inputs(name)
        outs = ref(inp["x"], inp["times"], inp["cond"], inp["mask"])
        PC.loss_of(outs, inp["gys"]).backward()
        grads = {k: p.grad for k, p in ref.named_parameters()}
        assert all(g is not None for g in grads.values())
        small = [k for k, g in grads.items() if g.numel() <= 256][:24]
        blob = {
            "case": name,
            "torch": torch.__version__,
            "outputs": [o.detach().clone() for o in PC.as_list(outs)],
            "grad_norm": {k: float(g.double().norm()) for k, g in grads.items()},
            "grad_probe": {k: float((g.double() * PC.probe_for(k, g.shape)).sum()) for k, g in grads.items()},
            "grad_full": {k: grads[k].clone() for k in small},
            "param_sum": {k: float(v.double().sum()) for k, v in sd.items()},
        }
        path = os.path.join(out_dir, name + ".pt")
        torch.save(blob, path)
        print("wrote", path, os.path.getsize(path), "bytes;", len(grads), "params; out mean|.| =",
              [float(o.abs().mean()) for o in blob["outputs"]])


if __name__ == "__main__":
    main()

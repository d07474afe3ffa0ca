"""Import the *real* reference (apple/ml-mdm at /root/reference) on CPU -- BUILD CONTAINER ONLY.

TEST INFRASTRUCTURE.  /root/reference does not exist on the GPU box; nothing that runs
there may import this module.  It is used by ``oracle/make_golden.py`` (fixture generation)
and by the ``reference``-marked CPU tests that pin ``oracle/unet_oracle.py``.

The reference's arithmetic is plain torch; only non-arithmetic imports are missing in
this image (torchinfo, simple_parsing, dataclass_wizard, mlx, torchvision, boto3 --
SURVEY.md section 8c).  They are replaced by empty stub modules.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MDM_REFERENCE_ROOT", "/root/reference/ml-mdm-matryoshka")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "ml_mdm"))


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


_loaded = None


def load():
    """-> namespace with .config .unet .nested_unet .diffusion .samplers .trainer .model_ema (reference modules)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # the reference tree is read-only
    if "torchinfo" not in sys.modules:
        _stub("torchinfo", summary=lambda *a, **k: None)
    if "simple_parsing" not in sys.modules:
        _stub("simple_parsing", ArgumentParser=object)
        _stub("simple_parsing.wrappers")
        _stub("simple_parsing.wrappers.field_wrapper", ArgumentGenerationMode=types.SimpleNamespace(BOTH=0))
    if "dataclass_wizard" not in sys.modules:
        _stub("dataclass_wizard", YAMLWizard=object)
    if "mlx" not in sys.modules:
        mlx = _stub("mlx")
        mlx.__path__ = []
        mlx.data = _stub("mlx.data", Buffer=object, Stream=object)
        mlx.data.core = _stub("mlx.data.core", CharTrie=object)
        mlx.core = _stub("mlx.core", array=type("array", (), {}))
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.utils = _stub("torchvision.utils", save_image=lambda *a, **k: None, make_grid=None)
    if "boto3" not in sys.modules:
        b3 = _stub("boto3")
        b3.session = _stub("boto3.session")
        _stub("boto3.s3")
        _stub("boto3.s3.transfer", TransferConfig=object)
    if "torch.utils.tensorboard" not in sys.modules:   # trainer.py:10 (type annotations only)
        try:
            import torch.utils.tensorboard  # noqa: F401
        except Exception:
            import torch.utils

            torch.utils.tensorboard = _stub("torch.utils.tensorboard", SummaryWriter=object)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from ml_mdm import config, diffusion, samplers, trainer  # noqa: E402
    from ml_mdm.models import model_ema, nested_unet, unet  # noqa: E402

    _loaded = types.SimpleNamespace(config=config, unet=unet, nested_unet=nested_unet, diffusion=diffusion, samplers=samplers,
                                    trainer=trainer, model_ema=model_ema)
    return _loaded

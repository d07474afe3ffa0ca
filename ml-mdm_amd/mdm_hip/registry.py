"""Plug the HIP denoisers into an importable reference checkout.

The reference resolves models through four plain dict registries (ml_mdm/config.py:9-63):
``get_model(name)`` returns ``MODEL_REGISTRY[MODEL_CONFIG_REGISTRY[name]["model"]]`` and the CLIs
(clis/train_parallel.py:66-72, clis/generate_sample.py:63-77, clis/generate_batch.py:98-107)
construct ``cls(input_channels, output_channels, config)``.  ``install()`` overwrites the two
model entries -- ``"unet"`` and ``"nested_unet"`` -- with this package's classes and leaves the
reference's config dataclasses, pipelines, samplers, trainer and CLIs untouched, so
``train_parallel`` / ``generate_sample`` run unmodified on the MI355X kernels.
"""


def install(ml_mdm_config=None):
    """Returns the dict of replaced entries.  ``ml_mdm_config`` defaults to ``ml_mdm.config``."""
    from .nested_unet import NestedUNet
    from .unet import UNet

    if ml_mdm_config is None:
        from ml_mdm import config as ml_mdm_config  # the reference package must be importable
        from ml_mdm.models import nested_unet as _n, unet as _u  # noqa: F401  (runs the reference's own registration first)
    previous = {k: ml_mdm_config.MODEL_REGISTRY.get(k) for k in ("unet", "nested_unet")}
    ml_mdm_config.MODEL_REGISTRY["unet"] = UNet
    ml_mdm_config.MODEL_REGISTRY["nested_unet"] = NestedUNet
    return previous

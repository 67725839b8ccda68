"""Make the reference's Hydra ``_target_`` strings resolve to this package.

``install_as_vsrlab()`` registers ``vsrlab_amd`` (and its sub-packages) under the name ``vsrlab`` in
``sys.modules`` -- the reference maps ``vsrlab`` to its ``src/`` the same way (setup.py:6-7) -- so
``vsrlab.vsr.models.RealBasicVSR.realbasicvsr.RealBasicVSR`` (conf/train/model/basicvsr.yaml:1)
instantiates the HIP-backed module with no YAML change.  ``instantiate`` is a 10-line stand-in for
``hydra.utils.instantiate(cfg, _recursive_=False)`` (core/utils.py:138) for hosts without Hydra."""
import importlib
import sys


def install_as_vsrlab(force: bool = False) -> None:
    import vsrlab_amd
    if "vsrlab" in sys.modules and sys.modules["vsrlab"] is not vsrlab_amd and not force:
        raise RuntimeError("a different `vsrlab` package is already imported; pass force=True to shadow it")
    sys.modules["vsrlab"] = vsrlab_amd
    for sub in ("core", "core.modules", "core.modules.conv", "core.modules.upsampling", "core.losses", "vsr", "vsr.models",
                "vsr.models.RealBasicVSR", "vsr.models.RealBasicVSR.realbasicvsr", "vsr.models.RealBasicVSR.modules",
                "vsr.models.RealBasicVSR.modules.basicvsr", "vsr.models.RealBasicVSR.modules.spynet",
                "vsr.models.RealBasicVSR.modules.unet-discriminator", "vsr.models.VRT", "vsr.models.VRT.modules",
                "vsr.models.VRT.modules.spynet", "vsr.models.VRT.modules.window_attention", "vsr.models.VRT.modules.tmsa",
                "core.utils", "train_gan", "optical_flow",
                "optical_flow.models", "optical_flow.models.spynet"):
        sys.modules["vsrlab." + sub] = importlib.import_module("vsrlab_amd." + sub)


def instantiate(cfg: dict):
    """``{'_target_': 'pkg.mod.Class', **kwargs}`` -> ``Class(**kwargs)``."""
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)(**cfg)

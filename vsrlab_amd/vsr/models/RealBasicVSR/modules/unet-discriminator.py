"""The reference's module name (``vsrlab.vsr.models.RealBasicVSR.modules.unet-discriminator``, conf/train/gan.yaml:17;
importable with importlib only, which is how Hydra resolves it).  Implementation: ``unet_discriminator.py``."""
from .unet_discriminator import UNetDiscriminator  # noqa: F401

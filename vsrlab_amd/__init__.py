"""vsrlab_amd -- the BasicVSR forward/backward hot path of santurini/vsrlab, MI355X-native.

HIP kernels + C ABI in ``csrc/`` (built to ``lib/libvsrlab_hip.so``), ctypes binding in ``_lib``,
autograd glue in ``functional``, and the reference's nn.Module plugin surface under ``vsr/``,
``core/`` and ``optical_flow/`` (same dotted paths as ``vsrlab.*``)."""
from . import _lib  # noqa: F401
from .compat import install_as_vsrlab, instantiate  # noqa: F401

__version__ = "0.1.0"

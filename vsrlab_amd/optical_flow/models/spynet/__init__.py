"""``vsrlab.optical_flow.models.spynet`` plugin surface (reference
``src/optical_flow/models/spynet/__init__.py:1-2``), backed by the same HIP SPyNet kernels.

The reference's own ``SpyNet``/``BasicModule`` here cannot run (NameError at model.py:43-44,
SURVEY.md appendix A6), so what is offered under these names is the BasicVSR-path SPyNet."""
from ....vsr.models.RealBasicVSR.modules.spynet import Spynet as SpyNet, SpynetModule as BasicModule, flow_warp  # noqa: F401

"""``vsrlab.optical_flow.models.spynet`` plugin surface (reference ``src/optical_flow/models/spynet/__init__.py:1-2``).

The reference's own classes here (``model.py:9-80``: ``BasicModule(input_channels)``, ``SpyNet(units, k, return_levels)``) cannot
run -- ``BasicModule.forward`` uses ``vsrlab.optical_flow`` as a variable (NameError at model.py:43-44) and
the training-size table it needs (``spynet.config``) is not in the tree -- so there is nothing to pin a re-implementation of
their ``forward`` against ("parity unpinned").  What is offered under these names: the constructors and ``state_dict`` keys of
the reference (``units.{i}.module.{0,2,4,6,8}.{weight,bias}``, so its checkpoints load strictly) around the canonical SPyNet
forward (no final ReLU) of the HIP engine, which is pinned by the VRT tree's SpyNet golden."""
from typing import Sequence

import torch
import torch.nn as nn

from .... import functional as VF
from ....vsr.models.RealBasicVSR.modules.spynet import flow_warp  # noqa: F401


class BasicModule(nn.Module):
    def __init__(self, input_channels: int = 8):
        super().__init__()
        if input_channels != 8:
            raise NotImplementedError("the HIP SPyNet level takes the canonical 8 input channels")
        self.module = nn.Sequential(
            nn.Conv2d(input_channels, 32, kernel_size=7, padding=3, stride=1), nn.ReLU(),
            nn.Conv2d(32, 64, kernel_size=7, padding=3, stride=1), nn.ReLU(),
            nn.Conv2d(64, 32, kernel_size=7, padding=3, stride=1), nn.ReLU(),
            nn.Conv2d(32, 16, kernel_size=7, padding=3, stride=1), nn.ReLU(),
            nn.Conv2d(16, 2, kernel_size=7, padding=3, stride=1))

    def params(self):
        out = []
        for j in (0, 2, 4, 6, 8):
            out += [self.module[j].weight, self.module[j].bias]
        return out

    def forward(self, x):
        """(N,8,h,w) = [first frame | warped second frame | upsampled flow] -> flow residual (N,2,h,w)."""
        return VF.spynet_module_forward(x, self.params(), last_relu=False)


class SpyNet(nn.Module):
    def __init__(self, units: Sequence[BasicModule] = None, k: int = None, return_levels=[2, 3, 4]):
        super().__init__()
        if units is not None and k is not None:
            assert len(units) == k
        if units is None and k is None:
            raise ValueError('At least one argument (units or k) must be specified')
        self.units = nn.ModuleList(units if units is not None else [BasicModule() for _ in range(k)])
        if len(self.units) != 6:
            raise NotImplementedError("the HIP SPyNet engine runs the 6-level pyramid")
        self.levels = len(self.units) - 1
        self.return_levels = list(return_levels)
        self.register_buffer('mean', torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), persistent=False)
        self.register_buffer('std', torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1), persistent=False)

    def forward(self, frames, train: bool = False, limit_k: int = -1):
        if train or limit_k != -1:
            raise NotImplementedError("only the evaluation path (train=False, all levels) is on the HIP engine")
        params = []
        for u in self.units:
            params += u.params()
        flows = VF.spynet_levels(params + [self.mean, self.std], frames[0], frames[1], self.return_levels, last_relu=False)
        return flows[0] if len(flows) == 1 else flows

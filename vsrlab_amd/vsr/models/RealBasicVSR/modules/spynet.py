"""SPyNet + flow_warp of the BasicVSR path, HIP-backed.

Mirrors vsrlab ``src/vsr/models/RealBasicVSR/modules/spynet.py``: ``SpynetModule`` (:13-21),
``Spynet`` (:23-93, including the ReLU after the last conv of every level and the resize to a
multiple of 32), ``flow_warp`` (:95-106).  Same constructor arguments, buffers and state_dict keys."""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from ..... import functional as VF
from ....._order import spynet_keys
from .....core.modules.conv import ConvReLU

flow_warp = VF.flow_warp


class SpynetModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.basic_module = nn.Sequential(ConvReLU(8, 32, 7, 1, 3), ConvReLU(32, 64, 7, 1, 3), ConvReLU(64, 32, 7, 1, 3),
                                          ConvReLU(32, 16, 7, 1, 3), ConvReLU(16, 2, 7, 1, 3))

    def forward(self, x):
        params = []
        for m in self.basic_module:
            params += [m.conv[0].weight, m.conv[0].bias]
        return VF.spynet_module_forward(x, params, last_relu=True)


class Spynet(nn.Module):
    def __init__(self, pretrained: bool = False):
        super().__init__()
        self.basic_module = nn.ModuleList([SpynetModule() for _ in range(6)])
        self.register_buffer('mean', torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer('std', torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))
        if pretrained:
            # same file and key remap as the reference (spynet.py:32-36); PROJECT_ROOT as in core/__init__.py
            root = os.environ.get("PROJECT_ROOT", os.getcwd())
            path = f'{root}/src/optical_flow/weights/spynet-sintel.pth'
            state_dict = torch.load(path, map_location="cpu", weights_only=True)
            new_dict = OrderedDict([(key[13:34] + '.0' + key[34:], state_dict[key]) for key in state_dict.keys()])
            self.basic_module.load_state_dict(new_dict)

    def ordered_tensors(self):
        """The 62 tensors in the C ABI's order (conv weights/biases level by level, then mean, std)."""
        sd = self.state_dict(keep_vars=True)
        return [sd[k] for k in spynet_keys()]

    def forward(self, ref, supp):
        return VF.spynet_flow(self.ordered_tensors(), ref, supp)

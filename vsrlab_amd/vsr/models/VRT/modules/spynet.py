"""``SpyNet`` of the VRT tree (vsrlab ``src/vsr/models/VRT/modules/spynet.py:68-157``; ``_target_`` of
``conf/train/model/spynet.yaml``) on the HIP SPyNet engine: the canonical network (NO ReLU after a level's last conv, unlike
the RealBasicVSR copy), ``return_levels`` multi-resolution flows, same constructor and ``state_dict`` keys
(``basic_module.{l}.basic_module.{0,2,4,6,8}.{weight,bias}``, ``mean``, ``std``).  Differentiable (parameters and both
frames, from the cotangents of every returned level); frames whose size is not a multiple of 32 are resized like the
reference does."""
import logging
import os

import torch
import torch.nn as nn

from ..... import functional as VF

pylogger = logging.getLogger(__name__)
flow_warp = VF.flow_warp


class BasicModule(nn.Module):
    """Basic Module for SpyNet (:68-82): five 7x7 convs, ReLU between them (none after the last)."""

    def __init__(self):
        super().__init__()
        self.basic_module = nn.Sequential(
            nn.Conv2d(in_channels=8, out_channels=32, kernel_size=7, stride=1, padding=3), nn.ReLU(inplace=False),
            nn.Conv2d(in_channels=32, out_channels=64, kernel_size=7, stride=1, padding=3), nn.ReLU(inplace=False),
            nn.Conv2d(in_channels=64, out_channels=32, kernel_size=7, stride=1, padding=3), nn.ReLU(inplace=False),
            nn.Conv2d(in_channels=32, out_channels=16, kernel_size=7, stride=1, padding=3), nn.ReLU(inplace=False),
            nn.Conv2d(in_channels=16, out_channels=2, kernel_size=7, stride=1, padding=3))

    def forward(self, tensor_input):
        raise NotImplementedError("BasicModule runs inside the SPyNet engine; call SpyNet(ref, supp)")


class SpyNet(nn.Module):
    def __init__(self, pretrained=True, return_levels=[5]):
        super().__init__()
        self.return_levels = list(return_levels)
        self.basic_module = nn.ModuleList([BasicModule() for _ in range(6)])
        if pretrained:
            pylogger.info('Loading Spynet pretrained weights')
            root = os.environ.get("PROJECT_ROOT", os.getcwd())
            load_path = f'{root}/src/vsr/models/VRT/weights/spynet_sintel_final-3d2a1287.pth'   # (the reference's path has a typo)
            self.load_state_dict(torch.load(load_path, map_location="cpu", weights_only=True)['params'])
        self.register_buffer('mean', torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer('std', torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def ordered_tensors(self):
        sd = self.state_dict(keep_vars=True)
        keys = []
        for lvl in range(6):
            for j in (0, 2, 4, 6, 8):
                keys += [f"basic_module.{lvl}.basic_module.{j}.weight", f"basic_module.{lvl}.basic_module.{j}.bias"]
        return [sd[k] for k in keys + ["mean", "std"]]

    def forward(self, ref, supp):
        assert ref.size() == supp.size()
        flows = VF.spynet_levels(self.ordered_tensors(), ref, supp, self.return_levels, last_relu=False)
        return flows[0] if len(flows) == 1 else flows

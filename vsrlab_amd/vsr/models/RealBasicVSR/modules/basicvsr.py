"""``BasicVSR`` on the MI355X engine.

Drop-in for vsrlab ``src/vsr/models/RealBasicVSR/modules/basicvsr.py:11-83``: same constructor
(``mid_channels=64, res_blocks=30, upscale=4, pretrained_flow=False, train_flow=False``), same
``state_dict`` keys/shapes, same call ``sr = model(lrs)`` with ``lrs`` (n,t,3,h,w) in [0,1] and
``sr`` (n,t,3,4h,4w).  Forward and backward of the whole clip are two calls into
libvsrlab_hip.so (``vsr_basicvsr_forward`` / ``vsr_basicvsr_backward``)."""
import logging
from typing import Optional

import torch
import torch.nn as nn

from ..... import functional as VF
from ....._order import basicvsr_keys
from .....core.modules.conv import ResidualBlock
from .....core.modules.upsampling import PixelShufflePack
from .spynet import Spynet, flow_warp  # noqa: F401  (flow_warp re-exported like the reference)

pylogger = logging.getLogger(__name__)


class BasicVSR(nn.Module):
    def __init__(self, mid_channels=64, res_blocks=30, upscale=4, pretrained_flow=False, train_flow=False):
        super().__init__()
        self.mid_channels = mid_channels
        self.backward_resblocks = ResidualBlock(mid_channels + 3, mid_channels, res_blocks)
        self.forward_resblocks = ResidualBlock(mid_channels + 3, mid_channels, res_blocks)
        self.point_conv = nn.Sequential(nn.Conv2d(mid_channels * 2, mid_channels, 1, 1), nn.LeakyReLU(0.1))
        self.upsample = nn.Sequential(*[PixelShufflePack(mid_channels, mid_channels, 2) for _ in range(upscale // 2)])
        self.conv_last = nn.Sequential(nn.Conv2d(mid_channels, 64, 3, 1, 1), nn.LeakyReLU(0.1), nn.Conv2d(64, 3, 3, 1, 1))
        self.upscale = nn.Upsample(scale_factor=upscale, mode='bilinear', align_corners=False)
        self.spynet = Spynet(pretrained_flow)
        self.res_blocks = res_blocks
        self.upscale_factor = upscale
        self.train_flow = train_flow
        #: 'fp32' | 'bf16' | None (= bf16 under autocast, else fp32; $VSRLAB_AMD_DTYPE overrides)
        self.compute_dtype: Optional[str] = None
        self._pool = VF.WorkspacePool()
        if not train_flow:
            pylogger.info('Setting Optical Flow weights to no_grad')
            for param in self.spynet.parameters():
                param.requires_grad = False

    # engine parameter order: include/vsrlab_hip.h / _order.py
    def _ordered_tensors(self):
        sd = self.state_dict(keep_vars=True)
        keys, n_trainable = basicvsr_keys(self.res_blocks, self.upscale_factor)
        return [sd[k] for k in keys], n_trainable

    def compute_flow(self, lrs):
        """(flow_forward, flow_backward), each (n*(t-1),2,h,w): the reference's helper (basicvsr.py:30-37).  ``forward`` does
        not call it (the engine computes the flows itself, readable with ``functional.basicvsr_flows``); here the frame
        pairs go through the HIP SPyNet in one batched call per direction."""
        n, t, c, h, w = lrs.size()
        earlier, later = lrs[:, :-1].reshape(-1, c, h, w), lrs[:, 1:].reshape(-1, c, h, w)
        return self.spynet(later, earlier), self.spynet(earlier, later)

    def forward(self, lrs):
        tensors, n_trainable = self._ordered_tensors()
        return VF.basicvsr_forward(lrs, tensors, n_trainable, self.mid_channels, self.res_blocks, self.upscale_factor,
                                   self._pool, self.compute_dtype)

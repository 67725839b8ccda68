"""``RealBasicVSR`` plugin surface (vsrlab ``src/vsr/models/RealBasicVSR/realbasicvsr.py:5-30``,
the ``_target_`` of conf/train/model/basicvsr.yaml).  ``sr, lq = model(lr)``.

Both stages run on the HIP engine, forward and backward: the pre-clean stack reuses the trunk kernels (3->64
stem on the planar frames, 64->64 residual blocks, 64->3 conv with the ``x +`` fused as a planar residual) and
its backward receives the gradient w.r.t. ``lq`` from BasicVSR (bilinear x4 skip, the stems' LR channels, the
flows through SPyNet's image pyramid).  ``lq`` is a fresh tensor; the reference mutates ``lr`` in place and
returns it (realbasicvsr.py:26-30) -- the values are identical, and the reference's own
backward only runs under autocast because of it."""
import torch.nn as nn

from .... import functional as VF
from ....core.modules.conv import ResidualBlock
from .modules.basicvsr import BasicVSR


class IterativeRefinement(nn.Module):
    def __init__(self, mid_ch, blocks, steps=3):
        super().__init__()
        self.steps = steps
        self.mid_ch = mid_ch
        self.blocks = blocks
        self.resblock = ResidualBlock(3, mid_ch, blocks)
        self.conv = nn.Conv2d(mid_ch, 3, 3, 1, 1, bias=True)

    def _ordered_tensors(self):
        sd = self.state_dict(keep_vars=True)
        keys = ["resblock.conv.0.weight", "resblock.conv.0.bias"]
        for i in range(self.blocks):
            for j in (1, 2):
                keys += [f"resblock.res_block.{i}.conv{j}.weight", f"resblock.res_block.{i}.conv{j}.bias"]
        return [sd[k] for k in keys + ["conv.weight", "conv.bias"]]

    def forward(self, x):
        return VF.cleaner_forward(self._ordered_tensors(), x, self.mid_ch, self.blocks, self.steps)


class RealBasicVSR(nn.Module):
    def __init__(self, cleaning_blocks=20, *args, **kwargs):
        super().__init__()
        self.cleaner = IterativeRefinement(kwargs["mid_channels"], cleaning_blocks)
        self.basicvsr = BasicVSR(*args, **kwargs)

    def forward(self, lr):
        lr = self.cleaner(lr)
        sr = self.basicvsr(lr)
        return sr, lr

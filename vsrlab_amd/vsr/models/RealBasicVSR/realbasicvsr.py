"""``RealBasicVSR`` plugin surface (vsrlab ``src/vsr/models/RealBasicVSR/realbasicvsr.py:5-30``,
the ``_target_`` of conf/train/model/basicvsr.yaml).  ``sr, lq = model(lr)``.

The BasicVSR stage runs on the HIP engine.  The pre-clean stack (``IterativeRefinement``) keeps the
reference's parameters and keys; its HIP kernels are the next row of the scope table,
so a non-zero ``cleaning_blocks`` forward raises instead of silently falling back."""
import torch.nn as nn

from ....core.modules.conv import ResidualBlock
from .modules.basicvsr import BasicVSR


class IterativeRefinement(nn.Module):
    def __init__(self, mid_ch, blocks, steps=3):
        super().__init__()
        self.steps = steps
        self.resblock = ResidualBlock(3, mid_ch, blocks)
        self.conv = nn.Conv2d(mid_ch, 3, 3, 1, 1, bias=True)

    def forward(self, x):
        raise NotImplementedError("the RealBasicVSR pre-clean stack is not on the HIP path yet (SURVEY.md 8f rank 1)")


class RealBasicVSR(nn.Module):
    def __init__(self, cleaning_blocks=20, *args, **kwargs):
        super().__init__()
        self.cleaner = IterativeRefinement(kwargs["mid_channels"], cleaning_blocks)
        self.basicvsr = BasicVSR(*args, **kwargs)

    def forward(self, lr):
        lr = self.cleaner(lr)
        sr = self.basicvsr(lr)
        return sr, lr

"""``CharbonnierLoss`` (vsrlab ``src/core/losses.py:10-18``), fused value+gradient HIP kernel."""
import torch.nn as nn

from .. import functional as VF


class CharbonnierLoss(nn.Module):
    def __init__(self, eps=1e-9):
        super().__init__()
        self.eps = eps

    def forward(self, x, y):
        return VF.charbonnier_loss(x, y, self.eps)

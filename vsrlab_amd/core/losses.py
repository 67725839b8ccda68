"""``CharbonnierLoss`` (vsrlab ``src/core/losses.py:10-18``), fused value+gradient HIP kernel."""
import torch.nn as nn

from .. import functional as VF


class CharbonnierLoss(nn.Module):
    def __init__(self, eps=1e-9):
        super().__init__()
        self.eps = eps

    def forward(self, x, y):
        return VF.charbonnier_loss(x, y, self.eps)


class AdversarialLoss(nn.Module):
    """``AdversarialLoss`` (``src/core/losses.py:66-74``): BCE-with-logits against a constant target map; scaled by
    ``weight`` on the generator side only.  Fused value + gradient HIP kernel."""

    def __init__(self, weight=2e-5):
        super().__init__()
        self.weight = weight

    def forward(self, x, target, is_disc=False):
        loss = VF.bce_with_logits_const(x, target)
        return loss if is_disc else loss * self.weight

"""Shared conv blocks of the hot path, HIP-backed.  Mirrors vsrlab ``src/core/modules/conv.py``:
``ConvReLU`` (:15-22), ``ResidualConv`` (:82-92), ``ResidualBlock`` (:94-103) -- same constructor
arguments, same parameter names (``conv.0.weight``, ``res_block.{i}.conv1.weight`` ...), so
checkpoints load strictly.  The modules are parameter containers: inside ``BasicVSR`` the whole
propagation runs in one engine call; called on their own they dispatch to the per-op kernels."""
import torch.nn as nn

from ... import functional as VF


class ConvReLU(nn.Module):
    """Conv2d + ReLU (SPyNet building block, conv.py:15-22)."""

    def __init__(self, in_ch, out_ch, *args, **kwargs):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_ch, out_ch, *args, **kwargs), nn.ReLU())

    def forward(self, x):
        raise NotImplementedError("ConvReLU is fused into the SPyNet engine; call Spynet(ref, supp)")


class ResidualConv(nn.Module):
    """x + conv2(relu(conv1(x)))  (conv.py:82-92)."""

    def __init__(self, filters=64):
        super().__init__()
        self.conv1 = nn.Conv2d(filters, filters, 3, 1, 1)
        self.conv2 = nn.Conv2d(filters, filters, 3, 1, 1)
        self.relu = nn.ReLU()

    def forward(self, x):
        return VF.residual_conv(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias)


class ResidualBlock(nn.Module):
    """conv3x3 + LeakyReLU(0.1), then ``blocks`` x ResidualConv  (conv.py:94-103)."""

    def __init__(self, in_ch, out_ch=64, blocks=30):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_ch, out_ch, 3, 1, 1), nn.LeakyReLU(0.1))
        self.res_block = nn.Sequential(*[ResidualConv(out_ch) for _ in range(blocks)])

    def forward(self, x):
        raise NotImplementedError("ResidualBlock runs inside the BasicVSR engine (stem conv on cat(lr, feat) is fused "
                                  "with the propagation); standalone use is not on the HIP path yet")

"""Shared conv blocks of the hot path, HIP-backed.  Mirrors vsrlab ``src/core/modules/conv.py``:
``ConvReLU`` (:15-22), ``ResidualConv`` (:82-92), ``ResidualBlock`` (:94-103) -- same constructor
arguments, same parameter names (``conv.0.weight``, ``res_block.{i}.conv1.weight`` ...), so
checkpoints load strictly.  Inside ``BasicVSR`` / ``Spynet`` / ``UNetDiscriminator`` the whole network runs in one engine
call and these modules are its parameter containers; called on their own they dispatch to the per-layer kernels
(all differentiable: ``ResidualConv`` through its fused forward/backward function, ``ConvReLU`` and the ``ResidualBlock`` stem through
``vsr_conv_layer_fwd`` / ``vsr_conv_layer_bwd``)."""
import torch
import torch.nn as nn

from ... import functional as VF


class _SpectralConv2d(nn.Module):
    """The parameters / buffers ``torch.nn.utils.spectral_norm(nn.Conv2d(..., bias=False))`` registers
    (``weight_orig``, ``weight_u``, ``weight_v``: same names, shapes and initialisation), so reference checkpoints
    load strictly.  The normalisation itself (one power iteration per training forward) runs in
    ``vsr_spectral_norm`` inside the discriminator's HIP call."""

    def __init__(self, in_ch, out_ch, ks, stride, pad):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding = in_ch, out_ch, ks, stride, pad
        conv = nn.Conv2d(in_ch, out_ch, ks, stride, pad, bias=False)          # for its default initialisation
        self.weight_orig = nn.Parameter(conv.weight.detach().clone())
        with torch.no_grad():
            u = nn.functional.normalize(torch.randn(out_ch), dim=0, eps=1e-12)
            v = nn.functional.normalize(torch.randn(in_ch * ks * ks), dim=0, eps=1e-12)
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)


class SpectralConv(nn.Module):
    """spectral_norm(Conv2d(in_ch, out_ch, ks, stride, pad, bias=False))  (conv.py:6-13).  A building block of
    ``UNetDiscriminator``, which runs all of its layers in one engine call; called on its own it is an ordinary differentiable
    module for the shape the per-layer kernels serve (64 -> 64, 3x3, stride 1, padding 1 = the class defaults at 64 channels):
    ``vsr_spectral_norm`` (one power iteration per training-mode forward, ``weight_u`` / ``weight_v`` updated in place) +
    ``vsr_conv_layer_fwd`` / ``_bwd``."""

    def __init__(self, in_ch, out_ch, ks=3, stride=1, pad=1):
        super().__init__()
        self.conv = _SpectralConv2d(in_ch, out_ch, ks, stride, pad)

    def forward(self, x):
        c = self.conv
        if (c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding) != (64, 64, 3, 1, 1):
            raise NotImplementedError("standalone HIP SpectralConv: Conv2d(64, 64, 3, 1, 1); the wide and the 4x4 stride-2 layers run "
                                      "inside the UNetDiscriminator engine (vsr_disc_forward)")
        return VF.spectral_conv_forward(x, c.weight_orig, c.weight_u, c.weight_v, self.training)


class ConvReLU(nn.Module):
    """Conv2d + ReLU (SPyNet building block, conv.py:15-22)."""

    def __init__(self, in_ch, out_ch, *args, **kwargs):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_ch, out_ch, *args, **kwargs), nn.ReLU())

    def forward(self, x):
        conv = self.conv[0]
        if conv.stride != (1, 1) or conv.padding != (conv.kernel_size[0] // 2,) * 2:
            raise NotImplementedError("HIP ConvReLU: stride 1, 'same' padding")
        return VF.conv_relu_forward(x, conv.weight, conv.bias)


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
        blocks = [(b.conv1.weight, b.conv1.bias, b.conv2.weight, b.conv2.bias) for b in self.res_block]
        return VF.residual_block_forward(x, self.conv[0].weight, self.conv[0].bias, blocks)

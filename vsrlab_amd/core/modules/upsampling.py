"""``PixelShufflePack`` (vsrlab ``src/core/modules/upsampling.py:4-12``): conv3x3 C->4C then
PixelShuffle(2), no activation.  On the HIP path the shuffle is the store pattern of the conv
kernel (four sub-convolutions, one launch).  Inside ``BasicVSR`` it runs in the engine; called on its own it is one
``vsr_conv_layer_fwd`` launch (forward-only)."""
import torch.nn as nn

from ... import functional as VF


class PixelShufflePack(nn.Module):
    def __init__(self, in_ch, out_ch, upscale_factor):
        super().__init__()
        self.upconv = nn.Conv2d(in_ch, out_ch * upscale_factor * upscale_factor, 3, 1, 1)
        self.pixel_shuffle = nn.PixelShuffle(upscale_factor)

    def forward(self, x):
        return VF.pixel_shuffle_pack_forward(x, self.upconv.weight, self.upconv.bias)

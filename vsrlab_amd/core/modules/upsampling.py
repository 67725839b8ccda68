"""``PixelShufflePack`` (vsrlab ``src/core/modules/upsampling.py:4-12``): conv3x3 C->4C then
PixelShuffle(2), no activation.  On the HIP path the shuffle is the store pattern of the conv
kernel (four sub-convolutions, one launch).  Inside ``BasicVSR`` it runs in the engine; called on its own it is one
``vsr_conv_layer_fwd`` launch; backward = ``vsr_conv_layer_bwd`` (four phase data-gradient launches + four weight-gradient launches)."""
import torch.nn as nn

from ... import functional as VF


class PixelShufflePack(nn.Module):
    def __init__(self, in_ch, out_ch, upscale_factor):
        super().__init__()
        r = int(upscale_factor)
        if r != 2:
            raise NotImplementedError("the HIP pixel-shuffle store pattern is built for upscale_factor 2 (BasicVSR x4 = two of them, basicvsr.py:19)")
        self.upconv = nn.Conv2d(in_ch, out_ch * r * r, kernel_size=3, stride=1, padding=1)     # key: upconv.{weight,bias}
        self.pixel_shuffle = nn.PixelShuffle(r)                                                # parameter-free; kept for attribute compatibility

    def forward(self, x):
        return VF.pixel_shuffle_pack_forward(x, self.upconv.weight, self.upconv.bias)

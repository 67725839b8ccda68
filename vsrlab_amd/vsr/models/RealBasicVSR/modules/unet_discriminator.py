"""``UNetDiscriminator`` on the MI355X engine (BASELINE config 3).

Drop-in for vsrlab ``src/vsr/models/RealBasicVSR/modules/unet-discriminator.py:4-31``: same constructor
(``in_ch=3, mid_ch=64``), same ``state_dict`` keys (``conv_0.{weight,bias}``, ``conv_k.conv.{weight_orig,weight_u,
weight_v}`` for k = 1..8, ``conv_9.{weight,bias}``), same call ``logits = D(img)`` with ``img`` (N,3,H,W).  Forward and
backward are one call each into libvsrlab_hip.so (``vsr_disc_forward`` / ``vsr_disc_backward``), with the spectral
normalisation of the eight inner convs (``vsr_spectral_norm``) in front.  The reference file name has a hyphen
(``conf/train/gan.yaml:17``); ``unet-discriminator.py`` next to this file re-exports the class under that name."""
import torch.nn as nn

from ..... import functional as VF
from .....core.modules.conv import SpectralConv


_SPECTRAL_LAYERS = ((1, 1, 2, 4, 2), (2, 2, 4, 4, 2), (3, 4, 8, 4, 2), (4, 8, 4, 3, 1), (5, 4, 2, 3, 1), (6, 2, 1, 3, 1), (7, 1, 1, 3, 1), (8, 1, 1, 3, 1))


class UNetDiscriminator(nn.Module):
    def __init__(self, in_ch=3, mid_ch=64):
        super().__init__()
        if in_ch != 3 or mid_ch != 64:
            raise NotImplementedError("the HIP discriminator is built for in_ch=3, mid_ch=64 (conf/train/gan.yaml:18-19)")
        # parameter containers only (the arithmetic is in csrc/disc_engine.hip); attribute names = the reference's state_dict keys.
        # (k, in multiple, out multiple, kernel, stride) of the eight spectrally normalised convs: three 4x4 stride-2 down, three 3x3 up, two 3x3 at 64
        self.conv_0 = nn.Conv2d(in_ch, mid_ch, 3, 1, 1)
        for k, cin, cout, ks, stride in _SPECTRAL_LAYERS:
            setattr(self, f"conv_{k}", SpectralConv(mid_ch * cin, mid_ch * cout, ks, stride, 1))
        self.conv_9 = nn.Conv2d(mid_ch, 1, 3, 1, 1)
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)     # parameter-free; kept for attribute compatibility
        self.lrelu = nn.LeakyReLU(0.2)
        #: 'fp32' | 'bf16' | None (= bf16 under autocast, else fp32; $VSRLAB_AMD_DTYPE overrides)
        self.compute_dtype = None

    def forward(self, img):
        sp = [getattr(self, f"conv_{k}").conv for k in range(1, 9)]
        params = [self.conv_0.weight, self.conv_0.bias] + [m.weight_orig for m in sp] + [self.conv_9.weight, self.conv_9.bias]
        bufs = []
        for m in sp:
            bufs += [m.weight_u, m.weight_v]
        return VF.discriminator_forward(img, params, bufs, self.training, self.compute_dtype)

"""CPU restatement of the GAN side of the reference's RealBasicVSR training (BASELINE config 3).

TEST INFRASTRUCTURE ONLY (like basicvsr_oracle.py): imported by tests/, __graft_entry__.smoke() and nothing else.
Functional, state_dict-keyed, fp32 or fp64; every function cites the reference lines it restates
(paths relative to /root/reference/src).  Pinned by tests/golden/unet_discriminator*.npz, generated from the imported
reference by tests/golden/make_golden.py (`disc`).
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from .basicvsr_oracle import _gq, _iq, _q, _wq, charbonnier     # emulate_bf16() rounding points (no-ops by default)


def disc_param_shapes(in_ch: int = 3, mid_ch: int = 64) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys -> shapes of ``UNetDiscriminator(in_ch, mid_ch)``
    (vsr/models/RealBasicVSR/modules/unet-discriminator.py:5-19; SpectralConv = spectral_norm(Conv2d(bias=False)),
    core/modules/conv.py:6-9: keys ``conv.weight_orig``, ``conv.weight_u`` (cout), ``conv.weight_v`` (cin*kh*kw))."""
    m = mid_ch
    spec = [(1, m, 2 * m, 4), (2, 2 * m, 4 * m, 4), (3, 4 * m, 8 * m, 4), (4, 8 * m, 4 * m, 3), (5, 4 * m, 2 * m, 3),
            (6, 2 * m, m, 3), (7, m, m, 3), (8, m, m, 3)]
    shapes: Dict[str, Tuple[int, ...]] = {"conv_0.weight": (m, in_ch, 3, 3), "conv_0.bias": (m,)}
    for k, ci, co, ks in spec:
        shapes[f"conv_{k}.conv.weight_orig"] = (co, ci, ks, ks)
        shapes[f"conv_{k}.conv.weight_u"] = (co,)
        shapes[f"conv_{k}.conv.weight_v"] = (ci * ks * ks,)
    shapes["conv_9.weight"] = (1, m, 3, 3)
    shapes["conv_9.bias"] = (1,)
    return shapes


def spectral_normalize(w_orig: Tensor, u: Tensor, v: Tensor, training: bool = True, eps: float = 1e-12):
    """torch.nn.utils.spectral_norm's hook (torch/nn/utils/spectral_norm.py ``compute_weight``, n_power_iterations=1,
    dim=0), which core/modules/conv.py:9 wraps every SpectralConv in.  Training: ONE power iteration under no_grad,
    v <- normalize(W^T u), u <- normalize(W v) (the buffers are updated in place), then sigma = u . (W v) WITH gradient
    through W only, weight = W_orig / sigma.  Returns (weight, u_new, v_new, sigma)."""
    wm = w_orig.reshape(w_orig.shape[0], -1)
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
    sigma = torch.dot(u, torch.mv(wm, v))
    return w_orig / sigma, u, v, sigma


def discriminator_forward(sd: Mapping[str, Tensor], img: Tensor, training: bool = True,
                          new_buffers: Optional[dict] = None) -> Tensor:
    """UNetDiscriminator.forward (unet-discriminator.py:21-31): img (N,3,H,W) -> logits (N,1,H,W); H, W multiples of 8.
    ``new_buffers`` (optional dict) receives the updated weight_u / weight_v of the training-mode power iteration."""
    lrelu = lambda x: F.leaky_relu(x, 0.2)                      # :19
    up = lambda x: F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)     # :18

    def sconv(k, x, stride, pad):
        pre = f"conv_{k}.conv."
        w, u, v, _ = spectral_normalize(sd[pre + "weight_orig"], sd[pre + "weight_u"], sd[pre + "weight_v"], training)
        if new_buffers is not None:
            new_buffers[pre + "weight_u"], new_buffers[pre + "weight_v"] = u.detach(), v.detach()
        return F.conv2d(x, _wq(w), None, stride, pad)

    # _q / _wq / _iq / _gq: identity unless basicvsr_oracle.emulate_bf16() is active; then they round where the HIP bf16
    # build stores a tensor (every activation written to HBM, the packed weights, the image staged into conv_0)
    f0 = _q(lrelu(F.conv2d(_iq(img), _wq(sd["conv_0.weight"]), sd["conv_0.bias"], 1, 1)))
    f1 = _q(lrelu(sconv(1, f0, 2, 1)))
    f2 = _q(lrelu(sconv(2, f1, 2, 1)))
    f3 = _q(lrelu(sconv(3, f2, 2, 1)))
    f3 = _q(up(f3))
    f4 = _q(up(_q(lrelu(sconv(4, f3, 1, 1))) + f2))
    f5 = _q(up(_q(lrelu(sconv(5, f4, 1, 1))) + f1))
    f6 = _q(_q(lrelu(sconv(6, f5, 1, 1))) + f0)
    out = _q(lrelu(sconv(7, f6, 1, 1)))
    out = _q(lrelu(sconv(8, out, 1, 1)))
    return _gq(F.conv2d(out, _wq(sd["conv_9.weight"]), sd["conv_9.bias"], 1, 1))


def adversarial_loss(x: Tensor, target: float, is_disc: bool = False, weight: float = 2e-5) -> Tensor:
    """AdversarialLoss.forward (core/losses.py:66-74): BCE-with-logits against a constant target map, mean reduction;
    scaled by ``weight`` only on the generator side."""
    loss = F.binary_cross_entropy_with_logits(x, torch.full_like(x, float(target)))
    return loss if is_disc else loss * weight


def resize_target(hr: Tensor, size) -> Tensor:
    """kornia.geometry.transform.resize(hr, (h, w)) as used by compute_loss (core/utils.py:239): bilinear,
    align_corners=False, antialias=False, over the trailing two dims."""
    lead = hr.shape[:-2]
    return F.interpolate(hr.reshape(-1, 1, *hr.shape[-2:]), size=tuple(size), mode="bilinear", align_corners=False).reshape(*lead, *size)


def compute_loss(sr: Tensor, hr: Tensor, lq: Optional[Tensor] = None) -> Tensor:
    """compute_loss with CharbonnierLoss (core/utils.py:235-240)."""
    loss = charbonnier(sr, hr)
    if lq is not None:
        loss = loss + charbonnier(lq, resize_target(hr, lq.shape[-2:]))
    return loss


def gan_losses(sd_d: Mapping[str, Tensor], sr: Tensor, hr: Tensor, lq: Optional[Tensor], adv_weight: float = 2e-5,
               training: bool = True):
    """The two losses of one GAN iteration with ``perceptual_loss: null`` (train_gan.py:35-58, :98 dummy_loss):
    generator: compute_loss(sr, hr, lq) + 0 + adversarial(D(sr), 1, False); discriminator: adversarial(D(hr), 1, True) +
    adversarial(D(sr.detach()), 0, True).  sr / hr: (b,t,3,H,W).  The reference runs D three times per iteration, each
    a training-mode forward (one power iteration each); this restatement threads the u / v buffers through in that order."""
    b, t, c, h, w = hr.shape
    bufs = dict(sd_d)
    nb: dict = {}
    d_sr = discriminator_forward(bufs, sr.reshape(-1, c, h, w), training, nb)
    bufs.update(nb)
    loss_g = compute_loss(sr, hr, lq) + adversarial_loss(d_sr, 1, False, adv_weight)
    d_hr = discriminator_forward(bufs, hr.reshape(-1, c, h, w), training, nb)
    bufs.update(nb)
    d_fake = discriminator_forward(bufs, sr.detach().reshape(-1, c, h, w), training, nb)
    bufs.update(nb)
    loss_d = adversarial_loss(d_hr, 1, True) + adversarial_loss(d_fake, 0, True)
    return loss_g, loss_d, {k: v for k, v in bufs.items() if k.endswith(("weight_u", "weight_v"))}

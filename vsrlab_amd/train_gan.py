"""The two losses of one GAN iteration of RealBasicVSR training, on the HIP path.

Mirrors the reference's ``src/train_gan.py:32-58`` (``dummy_loss``, ``generator_step``, ``discriminator_step``): same
signatures and return values, and the same ``autocast`` regions (train_gan.py:38-42,52-56): inside them the modules
pick the bf16 build (``functional.resolve_dtype``), so a caller that uses these functions as drop-ins gets the
documented bf16 path without setting ``compute_dtype`` / ``$VSRLAB_AMD_DTYPE`` (the reference's fp16 autocast maps to
bf16 storage / fp32 accumulate on MI355X; an enclosing ``torch.autocast(enabled=False)`` is NOT overridden by the
reference either).  The perceptual loss (VGG19 pretrained, core/losses.py:34) needs downloaded weights
and is out of scope: pass ``dummy_loss``, which is what the reference uses for ``perceptual_loss: null``
(train_gan.py:98)."""
import torch

from .core.utils import compute_loss


def _autocast():
    # the reference writes torch.cuda.amp.autocast() (fp16 + GradScaler); bf16 is its MI355X counterpart
    return torch.autocast("cuda", dtype=torch.bfloat16)


def dummy_loss(x, y):
    return torch.tensor(0, dtype=torch.float32, requires_grad=True)


def generator_step(model, discriminator, loss_fn, perceptual_loss, adversarial_loss, lr, hr):
    b, t, c, h, w = hr.shape
    with _autocast():
        sr, lq = model(lr)
        pixel_loss = compute_loss(loss_fn, sr, hr, lq)
        disc_sr = discriminator(sr.reshape(-1, c, h, w))
        perceptual_g = perceptual_loss(sr, hr)
    disc_fake_loss = adversarial_loss(disc_sr, 1, False)
    loss = pixel_loss + perceptual_g + disc_fake_loss
    return sr, loss, perceptual_g, disc_fake_loss


def discriminator_step(discriminator, adversarial_loss, sr, hr):
    b, t, c, h, w = hr.shape
    sr = sr.reshape(b * t, c, h, w)            # rearrange 'b t c h w -> (b t) c h w'
    hr = hr.reshape(b * t, c, h, w)
    with _autocast():
        disc_hr = discriminator(hr)
        disc_sr = discriminator(sr.detach())
    loss = adversarial_loss(disc_hr, 1, True) + adversarial_loss(disc_sr, 0, True)
    return loss

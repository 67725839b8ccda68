"""The pieces of the reference's ``src/core/utils.py`` that sit on the hot path's training step, on the HIP kernels:

* ``compute_loss`` (``core/utils.py:235-240``): ``loss_fn(sr, hr) + loss_fn(lq, resize(hr, (h, w)))`` -- the second term
  is the pre-clean stack's supervision (RealBasicVSR); ``resize`` is kornia's bilinear resize (``core/losses.py:4``:
  align_corners=False, no antialiasing), here one streaming kernel.
* ``update_weights`` (``core/utils.py:270-280``): same signature; with a ``vsrlab_amd.optim.FusedAdam`` the clip +
  step + zero_grad tail is two launches and a memset over flat arenas, and an optional ``grad_sync``
  (``vsrlab_amd.parallel.FlatGradSync``) exchanges gradients once per optimizer step (``no_sync`` semantics on the
  accumulation micro-steps).  With any other optimizer it does exactly what the reference does.

The harness around them (Hydra, loggers, data loading, checkpoint I/O) is out of scope.
"""
from __future__ import annotations

import torch

from .. import _lib
from ..functional import _f32c, _ptr, _require_gpu, _stream


def resize(x: torch.Tensor, size, interpolation: str = "bilinear", align_corners=None, antialias: bool = False) -> torch.Tensor:
    """``kornia.geometry.transform.resize(x, (h, w))`` for (..., H, W) tensors: bilinear, align_corners=False, no
    antialiasing (kornia's defaults, which ``compute_loss`` uses).  Not differentiated (its input is the HR target)."""
    if interpolation != "bilinear" or align_corners not in (None, False) or antialias:
        raise NotImplementedError("HIP resize implements kornia's default: bilinear, align_corners=False, antialias=False")
    _require_gpu(x)
    if x.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("HIP resize is not differentiated (compute_loss applies it to the target)")
    h, w = int(size[0]), int(size[1])
    H, W = x.shape[-2:]
    x32 = _f32c(x)
    out = torch.empty(x.shape[:-2] + (h, w), dtype=torch.float32, device=x.device)
    planes = x32.numel() // (H * W)
    lib = _lib.load()
    _lib.check(lib.vsr_resize_bilinear(_ptr(x32), _ptr(out), planes, H, W, h, w, _stream()), "resize_bilinear")
    return out


def compute_loss(loss_fn, sr, hr, lq=None):
    """Reference ``compute_loss`` (core/utils.py:235-240), verbatim semantics."""
    loss = loss_fn(sr, hr)
    if lq is not None:
        _, _, c, h, w = lq.size()
        loss = loss + loss_fn(lq, resize(hr, (h, w)))
    return loss


def update_weights(model, loss, scaler, scheduler, optimizer, num_grad_acc, grad_clip, i, grad_sync=None):
    """Reference ``update_weights`` (core/utils.py:270-280).  ``scaler`` may be None (bf16 needs no loss scaling;
    the reference's fp16 autocast does, train.py:74).  ``grad_sync``: optional ``FlatGradSync``."""
    from ..optim import FusedAdam
    loss = loss / num_grad_acc
    last = (i + 1) % num_grad_acc == 0
    if grad_sync is not None and not last:
        with grad_sync.no_sync():
            (scaler.scale(loss) if scaler is not None else loss).backward()
    else:
        (scaler.scale(loss) if scaler is not None else loss).backward()
    if not last:
        return
    if grad_sync is not None:
        # gradients that reached p.grad through autograd instead of the arena (discriminator, cleaner, attention parameters;
        # after nn.Module.zero_grad()'s set_to_none, or a user-assigned p.grad) must be IN the arena before it is exchanged:
        # otherwise the all-reduce ships stale arena contents and step() then steps each rank on its own local gradient
        if isinstance(optimizer, FusedAdam):
            optimizer.gather_foreign_grads()
        grad_sync.all_reduce()
    if isinstance(optimizer, FusedAdam):
        if scaler is not None and scaler.is_enabled():
            raise NotImplementedError("FusedAdam runs the bf16 / fp32 builds, which need no loss scaling: pass scaler=None or "
                                      "GradScaler(enabled=False) (the reference's GradScaler serves its fp16 autocast, train.py:74)")
        optimizer.step(max_grad_norm=grad_clip)                           # clip + Adam (skips the update on inf/nan)
    else:
        if scaler is not None:
            scaler.unscale_(optimizer)
        torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
        if scaler is not None:
            scaler.step(optimizer)
            scaler.update()
        else:
            optimizer.step()
    if scheduler is not None:
        scheduler.step()
    optimizer.zero_grad()

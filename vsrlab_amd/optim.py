"""Fused training-step glue for the HIP path.

Reference behaviour being replaced (``src/core/utils.py:270-280``, ``update_weights``)::

    scaler.unscale_(optimizer); clip_grad_norm_(model.parameters(), grad_clip)
    scaler.step(optimizer); scaler.update(); scheduler.step(); optimizer.zero_grad()

with ``torch.optim.Adam(lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0)`` (``conf/train/optimizer/adam.yaml``).

MI355X-first layout: parameters, gradients and both Adam moments of the model each live in ONE flat fp32
buffer (every tensor a 256-byte-aligned view of it).  The HIP backward accumulates its gradients straight into
the gradient arena (``functional._BasicVSRFn.backward``: no per-tensor ``AccumulateGrad`` add, no per-tensor
fill), ``zero_grad`` is one memset, the data-parallel exchange is one RCCL all-reduce over the arena
(``parallel.FlatGradSync``) and clip + Adam is one reduction launch + one elementwise launch
(``csrc/train_step.hip``) instead of ~10 foreach launches over 254-tensor lists.  Formulas are
``torch.optim.Adam``'s and ``clip_grad_norm_``'s; the GPU test pins them against torch to fp32 rounding.

There is no CPU fallback: parameters that are not on a GPU raise.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from . import _lib
from .functional import _ptr, _stream

_ALIGN = 64          # floats: 256-byte slots, like the gradient arena of functional.py


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` (amsgrad=False) + optional ``clip_grad_norm_`` in one fused HIP step.

    ``params``: an iterable of parameters (one group) -- e.g. ``model.parameters()``; frozen ones
    (``requires_grad=False``, the SPyNet of ``train_flow=False``) are skipped.  Construction re-points every
    parameter's ``.data`` into a flat buffer and gives it a ``.grad`` that is a view of the flat gradient arena, so
    it must happen AFTER ``model.to(device)``.  ``state_dict()`` has ``torch.optim.Adam``'s layout (per-parameter
    ``step`` / ``exp_avg`` / ``exp_avg_sq``), so the reference's checkpoints (``core/utils.py:83-88``) interchange.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-4, betas=(0.9, 0.99), eps: float = 1e-8,
                 weight_decay: float = 0.0, max_grad_norm: Optional[float] = None):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FusedAdam: no trainable parameters")
        for p in params:
            if not p.is_cuda:
                raise RuntimeError("vsrlab_amd.optim.FusedAdam runs on MI355X only (move the model to the GPU first); "
                                   "there is no CPU fallback")
            if p.dtype != torch.float32:
                raise TypeError("FusedAdam keeps fp32 master parameters (the HIP path's parameters are fp32)")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdam supports one parameter group (as the reference's configs)")
        self.max_grad_norm = max_grad_norm
        self._params = params
        dev = params[0].device
        self._offsets = []
        off = 0
        for p in params:
            self._offsets.append(off)
            off += ((p.numel() + _ALIGN - 1) // _ALIGN) * _ALIGN
        self._numel = off
        self.flat_params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        lib = _lib.load()
        self._scratch = torch.empty(lib.vsr_optim_scratch_floats(), dtype=torch.float32, device=dev)
        self.last_grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)   # total norm of the last step (device; no sync)
        self._step = 0
        self._step_t = torch.tensor(0.0)          # ONE host tensor shared by every parameter's state["step"]
        with torch.no_grad():
            for p, o in zip(params, self._offsets):
                n = p.numel()
                self.flat_params[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_params[o:o + n].view(p.shape)
                slot = self.flat_grads[o:o + n].view(p.shape)
                if p.grad is not None:
                    slot.copy_(p.grad)
                p.grad = slot
                p._vsr_grad_slot = slot            # functional._BasicVSRFn.backward accumulates into it directly
                self.state[p] = {"step": self._step_t, "exp_avg": self.flat_exp_avg[o:o + n].view(p.shape),
                                 "exp_avg_sq": self.flat_exp_avg_sq[o:o + n].view(p.shape)}

    # ------------------------------------------------------------------------------------------ #
    def zero_grad(self, set_to_none: bool = False):
        """One memset of the arena.  ``set_to_none`` is ignored: the ``.grad`` views are what the HIP backward writes."""
        self.flat_grads.zero_()
        for p in self._params:
            if p.grad is None or p.grad.data_ptr() != p._vsr_grad_slot.data_ptr():
                p.grad = p._vsr_grad_slot

    def gather_foreign_grads(self):
        """Copy into the arena every gradient that an op which does not know the arena left in ``p.grad`` (re-assigned by
        autograd after ``set_to_none``, or by the user), and re-point ``p.grad`` at its arena slot.  ``step()`` and
        ``grad_norm()`` call it; a data-parallel exchange of ``flat_grads`` must call it FIRST (``update_weights`` and
        ``FlatGradSync(optimizer=...)`` do)."""
        for p in self._params:
            slot = p._vsr_grad_slot
            if p.grad is None:
                p.grad = slot
            elif p.grad.data_ptr() != slot.data_ptr():
                slot.copy_(p.grad)
                p.grad = slot

    def grad_norm(self, grad_scale: float = 1.0) -> torch.Tensor:
        """``clip_grad_norm_``'s return value (global L2 norm of all gradients) as a 1-element device tensor."""
        self.gather_foreign_grads()
        out = torch.empty(1, dtype=torch.float32, device=self.flat_grads.device)
        lib = _lib.load()
        _lib.check(lib.vsr_grad_norm(_ptr(self.flat_grads), self._numel, float(grad_scale), _ptr(self._scratch), _ptr(out), _stream()),
                   "grad_norm")
        return out

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: Optional[float] = None, grad_scale: float = 1.0):
        """clip (``max_grad_norm`` or the constructor's; None = no clipping) + Adam, two launches.
        ``grad_scale`` multiplies the gradient first (1/num_grad_acc, 1/world_size after a SUM all-reduce, or a
        GradScaler's inverse scale); a non-finite gradient norm skips the update like ``GradScaler.step``.

        Step count after a skipped update: the kernel decides on the device whether the norm is finite, and this
        method never synchronises with it, so the host-side ``step`` (and ``state_dict()['state'][i]['step']``)
        advances on a skipped update as well, whereas ``GradScaler.step`` would not have called ``optimizer.step()``.
        The bias corrections of later steps are then those of step+1 (a factor within 1e-2 of 1 after ~100 steps at
        beta2 = 0.99).  bf16 / fp32 training without loss scaling does not produce inf/nan gradients in normal
        operation; ``last_grad_norm`` (device tensor) lets a caller detect a skipped step and call
        ``rewind_skipped_step()`` to restore torch's count.  Pinned by ``test_fused_adam_skipped_step_count``."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.gather_foreign_grads()
        g = self.param_groups[0]
        mx = self.max_grad_norm if max_grad_norm is None else max_grad_norm
        self._step += 1
        lib = _lib.load()
        _lib.check(lib.vsr_adam_clip_step(_ptr(self.flat_params), _ptr(self.flat_grads), _ptr(self.flat_exp_avg),
                                          _ptr(self.flat_exp_avg_sq), self._numel, float(g["lr"]), float(g["betas"][0]),
                                          float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._step,
                                          float(grad_scale), float(mx) if mx else 0.0, _ptr(self._scratch),
                                          _ptr(self.last_grad_norm), _stream()), "adam_clip_step")
        self._step_t.fill_(float(self._step))
        return loss

    def rewind_skipped_step(self):
        """If the last ``step()`` was skipped on the device (non-finite gradient norm), take its count back so that the step
        count matches what ``GradScaler.step`` + ``torch.optim.Adam`` would hold.  Synchronises (reads ``last_grad_norm``)."""
        if self._step > 0 and not bool(torch.isfinite(self.last_grad_norm).item()):
            from .functional import raise_on_chain_timeout
            raise_on_chain_timeout("FusedAdam skipped a step on a non-finite gradient norm")      # (synchronised already; raises if that is the cause)
            self._step -= 1
            self._step_t.fill_(float(self._step))
            return True
        return False

    def state_dict(self):
        """``torch.optim.Adam``'s layout.  A checkpoint is a host-side event: it also checks that no trunk-chain launch gave up a
        dependency wait since the last check (functional.raise_on_chain_timeout)."""
        from .functional import raise_on_chain_timeout
        raise_on_chain_timeout("FusedAdam.state_dict")
        return super().state_dict()

    # checkpoints: torch.optim.Adam's layout in, flat buffers refreshed --------------------------- #
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = set()
        with torch.no_grad():
            for p, o in zip(self._params, self._offsets):
                n = p.numel()
                st = self.state[p]
                for name, flat in (("exp_avg", self.flat_exp_avg), ("exp_avg_sq", self.flat_exp_avg_sq)):
                    view = flat[o:o + n].view(p.shape)
                    if name in st and st[name].data_ptr() != view.data_ptr():
                        view.copy_(st[name])
                    st[name] = view
                steps.add(int(float(st.get("step", 0.0))))
        if len(steps) != 1:
            raise ValueError("FusedAdam: all parameters share one step count")
        self._step = steps.pop()
        self._step_t = torch.tensor(float(self._step))
        for p in self._params:
            self.state[p]["step"] = self._step_t

"""Clip-level data parallelism for the HIP path: one process per GPU, one distinct batch of clips per
rank, ONE collective per optimizer step -- the sum of the flat gradient arena over RCCL / xGMI.

The reference wraps the model in ``DistributedDataParallel`` (``src/core/utils.py:147-151``), which still works with
these modules (gradients then flow through autograd so that DDP's bucket hooks fire).  This is the MI355X-first
form of the same exchange: the HIP backward leaves all gradients in the contiguous arena of
``optim.FusedAdam`` (19.4 MB for BasicVSR(64, 30)), so the all-reduce needs no buckets, no per-parameter hooks and
no gradient copies, and ``no_sync()`` during gradient accumulation (``conf/experiment/basic.yaml:26``,
``num_grad_acc: 4``: the reference all-reduces on every micro-step) costs nothing.  Results are identical to DDP:
the mean of the ranks' gradients.

Pure ``torch.distributed``: backend ``"nccl"`` is RCCL on ROCm; ``"gloo"`` runs the same code on CPU tensors (tests).
"""
from __future__ import annotations

import contextlib
from typing import Iterable, Optional

import torch
import torch.distributed as dist


class FlatGradSync:
    """``sync = FlatGradSync(optimizer.flat_grads, params=optimizer.flat_params)``; after ``loss.backward()``:
    ``sync.all_reduce()`` then ``optimizer.step()``.

    grads / params: flat tensors (views are fine).  ``optimizer``: the ``FusedAdam`` that owns the arena; gradients that
    autograd left outside the arena (``p.grad`` re-assigned after ``zero_grad(set_to_none=True)``) are then gathered into
    it before every exchange.  At construction the parameters are broadcast from rank 0 (DDP's
    start-up broadcast, so every replica starts from the same weights)."""

    def __init__(self, grads: torch.Tensor, params: Optional[torch.Tensor] = None, process_group=None,
                 extra_buffers: Iterable[torch.Tensor] = (), optimizer=None):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("FlatGradSync needs an initialised torch.distributed process group")
        self.grads = grads
        self.optimizer = optimizer               # a FusedAdam whose arena `grads` is: its foreign gradients are gathered before the exchange
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self._enabled = True
        self.num_collectives = 0                 # instrumentation for tests / logs
        if params is not None:
            dist.broadcast(params, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
        for b in extra_buffers:
            dist.broadcast(b, src=0, group=process_group)

    @contextlib.contextmanager
    def no_sync(self):
        """Skip the exchange inside the block (non-final gradient-accumulation micro-steps): the arena keeps
        accumulating locally and the next ``all_reduce()`` outside the block exchanges the sum."""
        prev, self._enabled = self._enabled, False
        try:
            yield
        finally:
            self._enabled = prev

    def all_reduce(self, async_op: bool = False):
        """grads <- mean over ranks.  Returns the work handle when ``async_op`` (call ``.wait()`` before the step)."""
        if not self._enabled or self.world_size == 1:
            return None
        if self.optimizer is not None:
            self.optimizer.gather_foreign_grads()
        self.num_collectives += 1
        work = dist.all_reduce(self.grads, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            return _ScaleAfter(work, self.grads, 1.0 / self.world_size)
        self.grads.mul_(1.0 / self.world_size)
        return None


class _ScaleAfter:
    def __init__(self, work, t, s):
        self.work, self.t, self.s = work, t, s

    def wait(self):
        self.work.wait()
        self.t.mul_(self.s)

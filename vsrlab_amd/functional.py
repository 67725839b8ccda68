"""torch-facing wrappers over the C ABI: PyTorch supplies device memory, the current HIP stream
and autograd bookkeeping; every FLOP runs in libvsrlab_hip.so.

There is deliberately no CPU / eager fallback: a tensor that is not on a GPU raises.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import BasicVSRDesc, DT_BF16, DT_F32

_DT = {"fp32": DT_F32, "float32": DT_F32, "f32": DT_F32, "bf16": DT_BF16, "bfloat16": DT_BF16}
_TORCH_DT = {DT_F32: torch.float32, DT_BF16: torch.bfloat16}


def resolve_dtype(compute_dtype: Optional[str] = None) -> int:
    """Internal activation dtype: explicit argument > $VSRLAB_AMD_DTYPE > bf16 under autocast > fp32.

    (The reference runs its forward under ``torch.cuda.amp.autocast()``, train.py:93; the
    MI355X build maps that to bf16 storage / fp32 accumulate.)"""
    name = compute_dtype or os.environ.get("VSRLAB_AMD_DTYPE")
    if name is None:
        name = "bf16" if torch.is_autocast_enabled("cuda") else "fp32"
    if name not in _DT:
        raise ValueError(f"unknown compute dtype {name!r}; use 'fp32' or 'bf16'")
    return _DT[name]


def _require_gpu(*tensors: torch.Tensor) -> None:
    for x in tensors:
        if not x.is_cuda:
            raise RuntimeError("vsrlab_amd runs on MI355X only: got a CPU tensor and there is no CPU fallback "
                               "(the CPU restatement lives in oracle/ and is test infrastructure)")


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(x: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if x is None else x.data_ptr())


def _ptr_array(tensors: Sequence[Optional[torch.Tensor]]):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, x in enumerate(tensors):
        arr[i] = 0 if x is None else x.data_ptr()
    return arr


def _f32c(x: torch.Tensor) -> torch.Tensor:
    return x.detach().to(torch.float32).contiguous()


# --------------------------------------------------------------------------------------------- #
# layout converters (reference planar fp32 <-> library pixel-major)
# --------------------------------------------------------------------------------------------- #
def to_pixel_major(x: torch.Tensor, dtype: int, channels: Optional[int] = None) -> torch.Tensor:
    """(N,C,H,W) fp32 planar -> the library's blocked pixel-major layout
    (N, H, ceil(W/32), Cp/8, 32, 8) in its element type (csrc/common.h).  The true width rides along as
    ``.pm_w`` (the padding pixels of the last 32-pixel segment are never read)."""
    _require_gpu(x)
    n, c, h, w = x.shape
    cp = channels or ((c + 15) // 16) * 16
    out = torch.empty((n, h, (w + 31) // 32, cp // 8, 32, 8), dtype=_TORCH_DT[dtype], device=x.device)
    out.pm_w = w
    lib = _lib.load()
    _lib.check(lib.vsr_planar_to_pm(dtype, _ptr(_f32c(x)), _ptr(out), n, c, h, w, cp, _stream()), "planar_to_pm")
    return out


def _pm_dims(x: torch.Tensor):
    n, h, ws, cp8, _, _ = x.shape
    return n, h, x.pm_w, cp8 * 8


def _pm_like(x: torch.Tensor, dtype=None) -> torch.Tensor:
    out = torch.empty_like(x) if dtype is None else torch.empty(x.shape, dtype=dtype, device=x.device)
    out.pm_w = x.pm_w
    return out


def from_pixel_major(x: torch.Tensor, channels: Optional[int] = None) -> torch.Tensor:
    """blocked pixel-major -> (N,C,H,W) fp32 planar."""
    _require_gpu(x)
    n, h, w, cp = _pm_dims(x)
    c = channels or cp
    dtype = DT_BF16 if x.dtype == torch.bfloat16 else DT_F32
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.vsr_pm_to_planar(dtype, _ptr(x), _ptr(out), n, c, h, w, cp, _stream()), "pm_to_planar")
    return out


# --------------------------------------------------------------------------------------------- #
# flow_warp  (reference: vsr/models/RealBasicVSR/modules/spynet.py:95-106)
# --------------------------------------------------------------------------------------------- #
class _FlowWarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flow_planar, dtype, border=0):
        n, c, h, w = x.shape
        lib = _lib.load()
        xin = to_pixel_major(x, dtype)
        cp = _pm_dims(xin)[3]
        out = _pm_like(xin)
        flow_planar = _f32c(flow_planar)
        _lib.check(lib.vsr_flow_warp_fwd_ex(dtype, _ptr(xin), _ptr(flow_planar), _ptr(out), n, h, w, cp, border, _stream()), "flow_warp_fwd")
        ctx.save_for_backward(flow_planar, xin if ctx.needs_input_grad[1] else None)
        ctx.meta = (dtype, c, cp, border)
        return from_pixel_major(out, c)

    @staticmethod
    def backward(ctx, gout):
        flow_planar, xin = ctx.saved_tensors
        dtype, c, cp, border = ctx.meta
        n, _, h, w = gout.shape
        lib = _lib.load()
        g = to_pixel_major(gout, dtype, cp)
        gx = gflow = None
        if ctx.needs_input_grad[0]:
            acc = torch.zeros((n, h, w, cp), dtype=torch.float32, device=gout.device)  # plain [N][H][W][C] fp32 accumulator
            _lib.check(lib.vsr_flow_warp_bwd_ex(dtype, _ptr(g), _ptr(flow_planar), _ptr(acc), n, h, w, cp, border, _stream()), "flow_warp_bwd")
            gx = acc[..., :c].permute(0, 3, 1, 2).contiguous()
        if ctx.needs_input_grad[1]:
            gflow = torch.empty((n, 2, h, w), dtype=torch.float32, device=gout.device)
            _lib.check(lib.vsr_flow_warp_bwd_flow_ex(dtype, _ptr(xin), _ptr(g), _ptr(flow_planar), _ptr(gflow), n, h, w, cp, border, _stream()),
                       "flow_warp_bwd_flow")
        return gx, gflow, None, None


def flow_warp(x: torch.Tensor, flow: torch.Tensor, interpolation: str = "bilinear", padding_mode: str = "zeros",
              align_corners: bool = True, compute_dtype: Optional[str] = None) -> torch.Tensor:
    """Drop-in for the reference ``flow_warp(x, flow)``: ``flow`` is channels-last (N,H,W,2),
    [...,0] = horizontal displacement in pixels.  Differentiable w.r.t. ``x`` and ``flow``."""
    if interpolation != "bilinear" or padding_mode not in ("zeros", "border") or not align_corners:
        raise NotImplementedError("HIP flow_warp implements the reference's two uses: bilinear, align_corners=True, padding "
                                  "'zeros' (propagation, basicvsr.py:55,70) or 'border' (SPyNet, spynet.py:60)")
    _require_gpu(x, flow)
    return _FlowWarpFn.apply(x, flow.permute(0, 3, 1, 2), resolve_dtype(compute_dtype), 1 if padding_mode == "border" else 0)


# --------------------------------------------------------------------------------------------- #
# conv3x3 64->64 pieces (reference: core/modules/conv.py:82-92) -- per-op entry points
# --------------------------------------------------------------------------------------------- #
def conv3x3_c64(x_pm: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0,
                res_pm: Optional[torch.Tensor] = None) -> torch.Tensor:
    _require_gpu(x_pm, weight)
    n, h, w, c = _pm_dims(x_pm)
    assert c == 64 and tuple(weight.shape) == (64, 64, 3, 3)
    dtype = DT_BF16 if x_pm.dtype == torch.bfloat16 else DT_F32
    lib = _lib.load()
    wpack = torch.empty(9 * 64 * 64, dtype=x_pm.dtype, device=x_pm.device)
    y = _pm_like(x_pm)
    _lib.check(lib.vsr_conv3x3_c64_fwd(dtype, _ptr(x_pm), _ptr(_f32c(weight)), _ptr(None if bias is None else _f32c(bias)),
                                       _ptr(wpack), _ptr(y), _ptr(res_pm), act, n, h, w, _stream()), "conv3x3_c64_fwd")
    return y


def chain_timeouts() -> int:
    """Dependency waits the trunk-chain kernel (csrc/conv3x3_chain.hip) has given up since the library was loaded (1 s each).
    Anything but 0 means some launch ran on unfinished inputs.  Such a launch poisons its own output with NaN on the device
    (chain_poison_kernel: sr, the loss and the gradient norm go non-finite, FusedAdam skips the step), so the counter is for
    diagnosis.  Synchronises the device."""
    import ctypes
    out = ctypes.c_uint(0)
    _lib.check(_lib.load().vsr_debug_chain_timeouts(ctypes.byref(out)), "debug_chain_timeouts")
    return int(out.value)


_chain_timeouts_seen = 0


def raise_on_chain_timeout(where: str = "") -> None:
    """For the places of a training loop that synchronise anyway (reading the loss, a checkpoint, `rewind_skipped_step`):
    raise if a chain launch has given up a dependency wait since the last call.  The launch concerned has already turned its
    outputs into NaN; this names the cause (a long foreign kernel, a debugger or another process holding the GPU's CUs)."""
    global _chain_timeouts_seen
    n = chain_timeouts()
    if n != _chain_timeouts_seen:
        new, _chain_timeouts_seen = n - _chain_timeouts_seen, n
        raise RuntimeError(f"vsrlab_amd: {new} dependency wait(s) of the trunk-chain kernel timed out{' (' + where + ')' if where else ''}: "
                           "the affected launches wrote NaN into their outputs; VSRLAB_AMD_CHAIN=0 runs one launch per layer")


class ResidualChainC64:
    """The body of ResidualBlock.res_block (conv.py:85-92, 99-103) as ONE launch (vsr_conv3x3_c64_chain_fwd): `blocks`
    ResidualConv blocks on a blocked 64-channel bf16 image, every intermediate kept.  Holds one allocation with the
    2*blocks + 1 images, the packed weights, the biases and the launch's flag words; call it with x_pm to run the chain,
    `image(l)` is layer l's input (image(2*blocks) = the output)."""

    def __init__(self, weights, biases, n: int, h: int, w: int, device):
        assert len(weights) == len(biases) and len(weights) >= 1
        lib = _lib.load()
        self.L, self.n, self.h, self.w = len(weights), n, h, w
        self.img_elems = n * h * ((w + 31) // 32) * 32 * 64
        al = lambda b: (b + 255) // 256 * 256
        o_w = al((self.L + 1) * self.img_elems * 2)
        o_b = o_w + al(self.L * 9 * 64 * 64 * 2)
        o_s = o_b + al(self.L * 64 * 4)
        nsync = lib.vsr_conv3x3_c64_chain_sync_bytes(self.L, n, h, w)
        if nsync == 0:
            raise ValueError("vsrlab_amd: unsupported chain shape")
        self.buf = torch.zeros(o_s + nsync + 512, dtype=torch.uint8, device=device)
        pad = (-self.buf.data_ptr()) % 256
        self.buf = self.buf[pad:]
        self.images = self.buf[:(self.L + 1) * self.img_elems * 2].view(torch.bfloat16)
        self.wpack = self.buf[o_w:o_w + self.L * 9 * 64 * 64 * 2].view(torch.bfloat16)
        self.bias = self.buf[o_b:o_b + self.L * 64 * 4].view(torch.float32)
        self.sync = self.buf[o_s:o_s + nsync]
        scratch = torch.empty(self.img_elems, dtype=torch.bfloat16, device=device)
        for l, (wt, b) in enumerate(zip(weights, biases)):          # vsr_conv3x3_c64_fwd packs a weight set as a side effect
            _lib.check(lib.vsr_conv3x3_c64_fwd(DT_BF16, _ptr(self.image(0)), _ptr(_f32c(wt)), _ptr(None), _ptr(self.wpack[l * 9 * 64 * 64:]),
                                               _ptr(scratch), _ptr(None), 0, n, h, w, _stream()), "conv3x3_c64_fwd")
            self.bias[l * 64:(l + 1) * 64] = 0.0 if b is None else _f32c(b)

    def image(self, l: int) -> torch.Tensor:
        return self.images[l * self.img_elems:(l + 1) * self.img_elems]

    def launch(self):
        lib = _lib.load()
        _lib.check(lib.vsr_conv3x3_c64_chain_fwd(_ptr(self.images), _ptr(self.wpack), _ptr(self.bias), self.L, self.n, self.h, self.w,
                                                 _ptr(self.sync), _stream()), "conv3x3_c64_chain_fwd")

    def __call__(self, x_pm: torch.Tensor) -> torch.Tensor:
        _require_gpu(x_pm)
        assert x_pm.dtype == torch.bfloat16 and x_pm.numel() == self.img_elems
        self.image(0).copy_(x_pm.reshape(-1))
        self.launch()
        return self.image(self.L)


def conv3x3_c64_dgrad(dy_pm: torch.Tensor, weight: torch.Tensor, res_pm: Optional[torch.Tensor] = None,
                      aux_pm: Optional[torch.Tensor] = None, mask_mode: int = 0) -> torch.Tensor:
    _require_gpu(dy_pm, weight)
    n, h, w, c = _pm_dims(dy_pm)
    dtype = DT_BF16 if dy_pm.dtype == torch.bfloat16 else DT_F32
    lib = _lib.load()
    wpack = torch.empty(9 * 64 * 64, dtype=dy_pm.dtype, device=dy_pm.device)
    dx = _pm_like(dy_pm)
    _lib.check(lib.vsr_conv3x3_c64_dgrad(dtype, _ptr(dy_pm), _ptr(_f32c(weight)), _ptr(wpack), _ptr(dx), _ptr(res_pm),
                                         _ptr(aux_pm), mask_mode, n, h, w, _stream()), "conv3x3_c64_dgrad")
    return dx


def conv3x3_c64_wgrad(x_pm: torch.Tensor, dy_pm: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    _require_gpu(x_pm, dy_pm)
    n, h, w, c = _pm_dims(x_pm)
    dtype = DT_BF16 if x_pm.dtype == torch.bfloat16 else DT_F32
    lib = _lib.load()
    slab = torch.empty(lib.vsr_conv3x3_c64_wgrad_slab_floats(), dtype=torch.float32, device=x_pm.device)
    gw = torch.empty((64, 64, 3, 3), dtype=torch.float32, device=x_pm.device)
    gb = torch.empty((64,), dtype=torch.float32, device=x_pm.device)
    _lib.check(lib.vsr_conv3x3_c64_wgrad(dtype, _ptr(x_pm), _ptr(dy_pm), _ptr(gw), _ptr(gb), _ptr(slab), n, h, w, _stream()),
               "conv3x3_c64_wgrad")
    return gw, gb


class _ResidualConvFn(torch.autograd.Function):
    """x + conv2(relu(conv1(x))) with its full backward, channels = 64 (conv.py:89-92)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, dtype):
        xp = to_pixel_major(x, dtype)
        a = conv3x3_c64(xp, w1, b1, act=1)
        y = conv3x3_c64(a, w2, b2, act=0, res_pm=xp)
        ctx.save_for_backward(xp, a, w1, w2)
        return from_pixel_major(y)

    @staticmethod
    def backward(ctx, gy):
        xp, a, w1, w2 = ctx.saved_tensors
        dtype = DT_BF16 if xp.dtype == torch.bfloat16 else DT_F32
        g = to_pixel_major(gy, dtype)
        ga = conv3x3_c64_dgrad(g, w2, aux_pm=a, mask_mode=1)
        gx = conv3x3_c64_dgrad(ga, w1, res_pm=g)
        gw2, gb2 = conv3x3_c64_wgrad(a, g)
        gw1, gb1 = conv3x3_c64_wgrad(xp, ga)
        return from_pixel_major(gx), gw1, gb1, gw2, gb2, None


def residual_conv(x, w1, b1, w2, b2, compute_dtype: Optional[str] = None):
    _require_gpu(x)
    if x.shape[1] != 64:
        raise NotImplementedError("the HIP ResidualConv is built for 64 channels (the reference default)")
    return _ResidualConvFn.apply(x, w1, b1, w2, b2, resolve_dtype(compute_dtype))


# --------------------------------------------------------------------------------------------- #
# SPyNet forward (reference: vsr/models/RealBasicVSR/modules/spynet.py:69-93)
# --------------------------------------------------------------------------------------------- #
class _SpynetFn(torch.autograd.Function):
    """flow = Spynet(ref, supp) with its backward on the HIP engine: parameter gradients (train_flow) and the gradient
    w.r.t. the two frames (border warps, image pyramid, /32 resize, normalisation)."""

    @staticmethod
    def forward(ctx, ref, supp, dtype, need_bwd, *params):
        n, _, h, w = ref.shape
        lib = _lib.load()
        ps = [_f32c(p) for p in params]
        ws = torch.empty(lib.vsr_spynet_workspace_bytes(n, h, w, dtype, int(need_bwd)), dtype=torch.uint8, device=ref.device)
        flow = torch.empty((n, 2, h, w), dtype=torch.float32, device=ref.device)
        _lib.check(lib.vsr_spynet_forward(n, h, w, dtype, _ptr_array(ps), len(ps), _ptr(_f32c(ref)), _ptr(_f32c(supp)), _ptr(flow),
                                          _ptr(ws), ws.numel(), int(need_bwd), _stream()), "spynet_forward")
        ctx.meta = (n, h, w, dtype, need_bwd)
        ctx.ws = ws if need_bwd else None
        ctx.ps = ps
        return flow

    @staticmethod
    def backward(ctx, dflow):
        n, h, w, dtype, need_bwd = ctx.meta
        if not need_bwd:
            raise RuntimeError("vsrlab_amd: backward through a SPyNet forward that ran without need_backward")
        if ctx.ws is None:
            raise RuntimeError("vsrlab_amd: trying to backward through the SPyNet graph a second time")
        lib = _lib.load()
        want_p = any(ctx.needs_input_grad[4 + k] for k in range(60))
        grads = [torch.zeros_like(p) if (k < 60 and want_p) else None for k, p in enumerate(ctx.ps)]
        dref = torch.empty((n, 3, h, w), dtype=torch.float32, device=dflow.device) if ctx.needs_input_grad[0] else None
        dsupp = torch.empty((n, 3, h, w), dtype=torch.float32, device=dflow.device) if ctx.needs_input_grad[1] else None
        _lib.check(lib.vsr_spynet_backward_ex(n, h, w, dtype, _ptr_array(ctx.ps), _ptr_array(grads) if want_p else None, len(grads),
                                              _ptr(_f32c(dflow)), 1, None, _ptr(dref), _ptr(dsupp), _ptr(ctx.ws), ctx.ws.numel(), _stream()),
                   "spynet_backward")
        ctx.ws = None
        return (dref, dsupp, None, None) + tuple(g if (g is not None and ctx.needs_input_grad[4 + k]) else None
                                                 for k, g in enumerate(grads))


def spynet_flow(params: Sequence[torch.Tensor], ref: torch.Tensor, supp: torch.Tensor,
                compute_dtype: Optional[str] = None) -> torch.Tensor:
    """``params``: the 62 Spynet tensors in state_dict order (60 conv tensors, mean, std).  Differentiable w.r.t. the
    60 conv tensors (train_flow, basicvsr.py:25-28) and w.r.t. both frames."""
    _require_gpu(ref, supp)
    if len(params) != 62:
        raise ValueError("expected the 62 tensors of Spynet.state_dict()")
    need_bwd = torch.is_grad_enabled() and (ref.requires_grad or supp.requires_grad or any(p.requires_grad for p in params[:60]))
    return _SpynetFn.apply(ref, supp, resolve_dtype(compute_dtype), need_bwd, *params)


class _SpynetLevelsFn(torch.autograd.Function):
    """The per-level flows of the canonical SPyNet with their backward on the HIP engine (vsr_spynet_forward_ex /
    vsr_spynet_backward_ex): parameter gradients and the gradient w.r.t. both frames, from the cotangents of every returned
    level (vsr/models/VRT/modules/spynet.py:98-157 under autograd)."""

    @staticmethod
    def forward(ctx, ref, supp, dtype, need_bwd, last_relu, levels, *params):
        n, _, h, w = ref.shape
        lib = _lib.load()
        ps = [_f32c(p) for p in params]
        ws = torch.empty(lib.vsr_spynet_workspace_bytes(n, h, w, dtype, int(need_bwd)), dtype=torch.uint8, device=ref.device)
        outs: List[Optional[torch.Tensor]] = [None] * 6
        for lv in levels:
            outs[lv] = torch.empty((n, 2, h >> (5 - lv), w >> (5 - lv)), dtype=torch.float32, device=ref.device)
        _lib.check(lib.vsr_spynet_forward_ex(n, h, w, dtype, _ptr_array(ps), len(ps), _ptr(_f32c(ref)), _ptr(_f32c(supp)), int(last_relu),
                                             _ptr_array(outs), _ptr(ws), ws.numel(), int(need_bwd), _stream()), "spynet_forward_ex")
        ctx.meta = (n, h, w, dtype, need_bwd, last_relu, levels)
        ctx.ws = ws if need_bwd else None
        ctx.ps = ps
        return tuple(outs[lv] for lv in levels)

    @staticmethod
    def backward(ctx, *douts):
        n, h, w, dtype, need_bwd, last_relu, levels = ctx.meta
        if not need_bwd:
            raise RuntimeError("vsrlab_amd: backward through a SPyNet forward that ran without need_backward")
        if ctx.ws is None:
            raise RuntimeError("vsrlab_amd: trying to backward through the SPyNet graph a second time")
        lib = _lib.load()
        dlevel: List[Optional[torch.Tensor]] = [None] * 6
        for lv, d in zip(levels, douts):
            if d is not None:
                dlevel[lv] = _f32c(d)
        want_p = any(ctx.needs_input_grad[6 + k] for k in range(60))
        grads = [torch.zeros_like(p) if (k < 60 and want_p) else None for k, p in enumerate(ctx.ps)]
        dev = ctx.ps[0].device
        dref = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dsupp = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        if all(d is None for d in dlevel):
            if dref is not None:
                dref.zero_()
            if dsupp is not None:
                dsupp.zero_()
        else:
            _lib.check(lib.vsr_spynet_backward_ex(n, h, w, dtype, _ptr_array(ctx.ps), _ptr_array(grads) if want_p else None, len(grads),
                                                  None, int(last_relu), _ptr_array(dlevel), _ptr(dref), _ptr(dsupp), _ptr(ctx.ws),
                                                  ctx.ws.numel(), _stream()), "spynet_backward")
        ctx.ws = None
        return (dref, dsupp, None, None, None, None) + tuple(g if (g is not None and ctx.needs_input_grad[6 + k]) else None
                                                             for k, g in enumerate(grads))


def spynet_levels(params: Sequence[torch.Tensor], ref: torch.Tensor, supp: torch.Tensor, return_levels: Sequence[int],
                  last_relu: bool = False, compute_dtype: Optional[str] = None) -> List[torch.Tensor]:
    """The canonical SPyNet (vsr/models/VRT/modules/spynet.py:68-157): the flows of the requested pyramid levels (5 = full
    resolution, 4 = 1/2, ...), finest first, like the reference's ``flow_list``; differentiable w.r.t. the 60 conv tensors
    and both frames."""
    _require_gpu(ref, supp)
    if len(params) != 62:
        raise ValueError("expected the 62 tensors of SpyNet.state_dict()")
    levels = tuple(sorted(set(int(v) for v in return_levels), reverse=True))
    if not levels or not all(0 <= lv <= 5 for lv in levels):
        raise ValueError("return_levels are pyramid levels 0..5")
    need_bwd = torch.is_grad_enabled() and (ref.requires_grad or supp.requires_grad or any(p.requires_grad for p in params[:60]))
    return list(_SpynetLevelsFn.apply(ref, supp, resolve_dtype(compute_dtype), need_bwd, bool(last_relu), levels, *params))


# --------------------------------------------------------------------------------------------- #
# whole BasicVSR path (reference: modules/basicvsr.py:39-83 + autograd)
# --------------------------------------------------------------------------------------------- #
def _under_ddp() -> bool:
    """True while a torch DistributedDataParallel wrapper is running this forward (DDP's own class-level
    ``_active_ddp_module`` marker, set for the duration of its forward)."""
    try:
        from torch.nn.parallel import DistributedDataParallel
        return getattr(DistributedDataParallel, "_active_ddp_module", None) is not None
    except Exception:
        return False


class Workspace:
    """One arena per forward, owned by that forward's autograd node (``ctx.ws``) until its backward has run or the graph is
    dropped, then returned to PyTorch's caching allocator -- which hands the same block back to the next forward of the same
    size without a hipMalloc, and to anybody else in between (round 2: the GAN iteration needs the generator's 87 GB for the
    discriminator's activations while the generator is idle; a private pool that kept the arena made config 3 run out of
    memory)."""

    def __init__(self, nbytes: int, device):
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.owner = None


class WorkspacePool:
    """Allocates arenas and remembers the most recent one WEAKLY (for ``basicvsr_flows``, which reads the flows out of a live
    forward's arena).  It does not keep arenas alive."""

    def __init__(self):
        self._last = None

    def acquire(self, key, nbytes: int, device) -> Workspace:
        ws = Workspace(nbytes, device)
        self._last = weakref.ref(ws)
        return ws

    def last(self) -> Optional[Workspace]:
        return self._last() if self._last is not None else None

    def clear(self):
        self._last = None


class _CtxToken:
    """weakref-able handle tying a workspace to the lifetime of an autograd graph node."""


class _BasicVSRFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lrs, desc_tuple, pool, n_trainable, need_bwd, direct, *params):
        n, t, h, w, mid, rb, up, dtype, arena = desc_tuple
        desc = BasicVSRDesc(n, t, h, w, mid, rb, up, dtype, arena)
        lib = _lib.load()
        nbytes = lib.vsr_basicvsr_workspace_bytes(ctypes.byref(desc), int(need_bwd))
        if nbytes == 0:
            raise RuntimeError(f"vsrlab_amd: unsupported BasicVSR configuration for the HIP path: {desc_tuple}")
        ws = pool.acquire((desc_tuple, need_bwd, lrs.device.index), nbytes, lrs.device)
        lr32 = _f32c(lrs)
        ps = [_f32c(p) for p in params]
        sr = torch.empty((n, t, 3, up * h, up * w), dtype=torch.float32, device=lrs.device)
        _lib.check(lib.vsr_basicvsr_forward(ctypes.byref(desc), _ptr_array(ps), len(ps), _ptr(lr32), _ptr(sr), _ptr(ws.buf),
                                            ws.buf.numel(), int(need_bwd), _stream()), "basicvsr_forward")
        if need_bwd:
            token = _CtxToken()
            ctx.token = token
            ws.owner = weakref.ref(token)
        ctx.ws = ws
        ctx.desc_tuple = desc_tuple
        ctx.n_trainable = n_trainable
        ctx.need_bwd = need_bwd
        ctx.lr32 = lr32
        ctx.ps = ps
        ctx.direct = direct
        ctx.params = params                 # the leaves themselves: their .grad is re-attached to the arena slot
        ctx.consumed = False
        return sr

    @staticmethod
    def backward(ctx, dsr):
        if not ctx.need_bwd:
            raise RuntimeError("vsrlab_amd: backward through a forward that ran without need_backward")
        if ctx.consumed:
            # the arena went back to the pool after the first backward (a later forward of the same shape may have
            # overwritten the saved activations): fail like PyTorch does for freed buffers instead of returning garbage
            raise RuntimeError("vsrlab_amd: trying to backward through the BasicVSR graph a second time: the saved "
                               "activations live in a pooled workspace that was released by the first backward")
        n, t, h, w, mid, rb, up, dtype, arena = ctx.desc_tuple
        desc = BasicVSRDesc(n, t, h, w, mid, rb, up, dtype, arena)
        lib = _lib.load()
        ps = ctx.ps
        # need_bwd == 2 (train_flow / input gradient): the SPyNet conv weights / biases (everything but the trailing mean,
        # std buffers) may be differentiated too
        n_diff = len(ps) - 2 if ctx.need_bwd == 2 else ctx.n_trainable
        # Tensors come in (weight, bias) pairs and the engine produces a pair's gradients together: a pair gets slots iff
        # one of its members needs a gradient; everything else is NULL (e.g. all of a frozen SPyNet when only the input
        # clip is differentiated: no 7x7 weight gradient is computed at all).
        need = [k < n_diff and bool(ctx.needs_input_grad[6 + k]) for k in range(len(ps))]
        want = [k < n_diff and (need[k & ~1] or need[k | 1]) for k in range(len(ps))]
        # Gradient destinations.  A parameter managed by vsrlab_amd.optim.FusedAdam carries `_vsr_grad_slot`, a view of
        # the optimizer's flat gradient arena: the engine accumulates (+=) straight into it -- what autograd's
        # AccumulateGrad would do -- and autograd gets None for it.  Everything else goes through ONE zero-filled
        # scratch arena (a fill kernel per tensor costs ~1 ms per step at 254 tensors) and is returned to autograd.
        direct = ctx.direct if ctx.direct is not None else [None] * len(ps)
        sizes = [ps[k].numel() if (want[k] and direct[k] is None) else 0 for k in range(len(ps))]
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + ((sz + 63) // 64) * 64)          # 256-byte aligned slots
        flat = torch.zeros(max(offs[-1], 1), dtype=torch.float32, device=dsr.device)
        dest: List[Optional[torch.Tensor]] = []
        for k, p in enumerate(ps):
            if not want[k]:
                dest.append(None)
            elif direct[k] is not None:
                dest.append(direct[k])
            else:
                dest.append(flat[offs[k]:offs[k] + sizes[k]].view(p.shape))
        dlrs = torch.empty_like(ctx.lr32) if ctx.needs_input_grad[0] else None      # gradient w.r.t. the clip (need_bwd == 2)
        _lib.check(lib.vsr_basicvsr_backward(ctypes.byref(desc), _ptr_array(ps), _ptr_array(dest), len(ps), _ptr(ctx.lr32),
                                             _ptr(_f32c(dsr)), _ptr(dlrs), _ptr(ctx.ws.buf), ctx.ws.buf.numel(), _stream()),
                   "basicvsr_backward")
        ctx.consumed = True
        ctx.ws.owner = None
        ctx.ws = None                    # the arena goes back to the allocator now, not when the caller drops the loss tensor
        ctx.token = None
        grads: List[Optional[torch.Tensor]] = []
        for k in range(len(ps)):
            if not need[k]:
                grads.append(None)
            elif direct[k] is not None:
                leaf = ctx.params[k]
                if leaf.grad is None or leaf.grad.data_ptr() != direct[k].data_ptr():
                    leaf.grad = direct[k]                            # e.g. after zero_grad(set_to_none=True)
                grads.append(None)
            else:
                grads.append(dest[k])
        ctx.params = None
        return (dlrs, None, None, None, None, None) + tuple(grads)


_ARENA_MODE: Optional[int] = None


def set_arena_mode(mode: Optional[str]) -> None:
    """Training workspace of the BasicVSR engine (VsrBasicVSRDesc.arena_mode, include/vsrlab_hip.h): "full" (default: every
    activation gradient of a direction kept, all-frames weight-gradient launches; 113 GiB per clip for BASELINE config 2) or "diet" (65 GiB: per-frame
    weight gradients behind a two-block gradient ring, HR activations recomputed; for batch-of-clips per GPU).  None: back to
    $VSRLAB_AMD_ARENA (unset = "full")."""
    global _ARENA_MODE
    if mode is not None and mode not in ("full", "diet"):
        raise ValueError("arena mode is 'full' or 'diet'")
    _ARENA_MODE = None if mode is None else int(mode == "diet")


def arena_mode() -> int:
    if _ARENA_MODE is not None:
        return _ARENA_MODE
    env = os.environ.get("VSRLAB_AMD_ARENA", "full")
    if env not in ("full", "diet"):
        raise ValueError("VSRLAB_AMD_ARENA is 'full' or 'diet'")
    return int(env == "diet")


def basicvsr_forward(lrs: torch.Tensor, params: Sequence[torch.Tensor], n_trainable: int, mid_channels: int,
                     res_blocks: int, upscale: int, pool: WorkspacePool, compute_dtype: Optional[str] = None) -> torch.Tensor:
    """sr = BasicVSR(lrs) on the HIP engine.  ``params`` in state_dict order; the first
    ``n_trainable`` are the non-SPyNet tensors.  SPyNet is frozen unless ``train_flow`` (basicvsr.py:25-28): if any
    of its conv parameters requires grad the engine also keeps SPyNet's activations and differentiates it
    (need_backward = 2)."""
    _require_gpu(lrs)
    if lrs.dim() != 5 or lrs.shape[2] != 3:
        raise ValueError("lrs must be (n,t,3,h,w)")
    n, t, _, h, w = lrs.shape
    desc_tuple = (n, t, h, w, mid_channels, res_blocks, upscale, resolve_dtype(compute_dtype), arena_mode())
    # grad mode is off inside Function.forward, so decide here whether activations must be retained
    need_bwd = 0
    if torch.is_grad_enabled():
        if lrs.requires_grad or (t > 1 and any(p.requires_grad for p in params[n_trainable:])):
            need_bwd = 2                                      # input-clip and / or SPyNet gradients: SPyNet's activations are kept
        elif any(p.requires_grad for p in params[:n_trainable]):
            need_bwd = 1
    # parameters owned by vsrlab_amd.optim.FusedAdam receive their gradient in place (see _BasicVSRFn.backward);
    # under DistributedDataParallel the gradients must flow through autograd so that DDP's hooks fire
    direct = [getattr(p, "_vsr_grad_slot", None) if (p.requires_grad and not _under_ddp()) else None for p in params]
    if not any(d is not None for d in direct):
        direct = None
    return _BasicVSRFn.apply(lrs, desc_tuple, pool, n_trainable, need_bwd, direct, *params)


def basicvsr_flows(lrs_shape, mid_channels, res_blocks, upscale, ws: Workspace, dtype: int, device):
    n, t, _, h, w = lrs_shape
    desc = BasicVSRDesc(n, t, h, w, mid_channels, res_blocks, upscale, dtype)
    ff = torch.empty((n, t - 1, 2, h, w), dtype=torch.float32, device=device)
    fb = torch.empty_like(ff)
    lib = _lib.load()
    _lib.check(lib.vsr_basicvsr_get_flows(ctypes.byref(desc), _ptr(ws.buf), _ptr(ff), _ptr(fb), _stream()), "get_flows")
    return ff, fb


# --------------------------------------------------------------------------------------------- #
# RealBasicVSR pre-clean stack, forward (reference: vsr/models/RealBasicVSR/realbasicvsr.py:17-30)
# --------------------------------------------------------------------------------------------- #
class _CleanerFn(torch.autograd.Function):
    """lq = IterativeRefinement(lr) with its backward on the HIP engine (parameter gradients and d lr)."""

    @staticmethod
    def forward(ctx, lr, meta, *params):
        mid_channels, blocks, steps, dtype, need_bwd = meta
        n, t, _, h, w = lr.shape
        lib = _lib.load()
        ps = [_f32c(p) for p in params]
        lr32 = _f32c(lr)
        nbytes = lib.vsr_cleaner_workspace_bytes(n * t, h, w, blocks, steps, dtype, int(need_bwd))
        if nbytes == 0:
            raise RuntimeError("vsrlab_amd: unsupported pre-clean configuration for the HIP path")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=lr.device)
        lq = torch.empty((n, t, 3, h, w), dtype=torch.float32, device=lr.device)
        _lib.check(lib.vsr_cleaner_forward(n * t, h, w, mid_channels, blocks, steps, dtype, _ptr_array(ps), len(ps), _ptr(lr32),
                                           _ptr(lq), _ptr(ws), ws.numel(), int(need_bwd), _stream()), "cleaner_forward")
        ctx.meta = meta
        ctx.ws = ws if need_bwd else None
        ctx.lr32 = lr32 if need_bwd else None
        ctx.shapes = [p.shape for p in ps]
        return lq

    @staticmethod
    def backward(ctx, dlq):
        mid_channels, blocks, steps, dtype, need_bwd = ctx.meta
        if not need_bwd:
            raise RuntimeError("vsrlab_amd: backward through a pre-clean forward that ran without need_backward")
        if ctx.ws is None:
            raise RuntimeError("vsrlab_amd: trying to backward through the pre-clean graph a second time")
        n, t, _, h, w = dlq.shape
        lib = _lib.load()
        grads = [torch.zeros(sh, dtype=torch.float32, device=dlq.device) for sh in ctx.shapes]
        dlr = torch.empty_like(ctx.lr32) if ctx.needs_input_grad[0] else None
        _lib.check(lib.vsr_cleaner_backward(n * t, h, w, mid_channels, blocks, steps, dtype, _ptr_array(grads), len(grads),
                                            _ptr(ctx.lr32), _ptr(_f32c(dlq)), _ptr(dlr), _ptr(ctx.ws), ctx.ws.numel(), _stream()),
                   "cleaner_backward")
        ctx.ws = None
        return (dlr, None) + tuple(g if ctx.needs_input_grad[2 + k] else None for k, g in enumerate(grads))


def cleaner_forward(params: Sequence[torch.Tensor], lr: torch.Tensor, mid_channels: int, blocks: int, steps: int = 3,
                    compute_dtype: Optional[str] = None) -> torch.Tensor:
    """lq = IterativeRefinement(lr): (n,t,3,h,w) -> (n,t,3,h,w), a fresh tensor (the reference adds in place,
    realbasicvsr.py:29).  Differentiable w.r.t. the parameters and lr."""
    _require_gpu(lr)
    if len(params) != 4 + 4 * blocks:
        raise ValueError("expected the 4 + 4*blocks tensors of IterativeRefinement")
    n, t, c, h, w = lr.shape
    if c != 3:
        raise ValueError("lr must be (n,t,3,h,w)")
    need_bwd = torch.is_grad_enabled() and (lr.requires_grad or any(p.requires_grad for p in params))
    return _CleanerFn.apply(lr, (mid_channels, blocks, steps, resolve_dtype(compute_dtype), need_bwd), *params)


# --------------------------------------------------------------------------------------------- #
# Charbonnier loss (reference: core/losses.py:10-18), fused value + gradient
# --------------------------------------------------------------------------------------------- #
class _CharbonnierFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, eps):
        lib = _lib.load()
        x32, y32 = _f32c(x), _f32c(y)
        dx = torch.empty_like(x32)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        scratch = torch.empty(lib.vsr_charbonnier_scratch_floats(), dtype=torch.float32, device=x.device)
        _lib.check(lib.vsr_charbonnier_fwd_bwd(_ptr(x32), _ptr(y32), _ptr(dx), _ptr(loss), _ptr(scratch), x32.numel(), float(eps), _stream()),
                   "charbonnier")
        ctx.save_for_backward(dx)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None, None


def charbonnier_loss(x: torch.Tensor, y: torch.Tensor, eps: float = 1e-9) -> torch.Tensor:
    _require_gpu(x, y)
    if y.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("HIP Charbonnier differentiates w.r.t. its first argument only")
    return _CharbonnierFn.apply(x, y, eps)


# --------------------------------------------------------------------------------------------- #
# GAN side (BASELINE config 3): UNetDiscriminator + spectral norm + BCE-with-logits
# (reference: vsr/models/RealBasicVSR/modules/unet-discriminator.py:4-31, core/modules/conv.py:6-13,
#  core/losses.py:66-74)
# --------------------------------------------------------------------------------------------- #
_DISC_SPECTRAL = [(128, 64, 4), (256, 128, 4), (512, 256, 4), (256, 512, 3), (128, 256, 3), (64, 128, 3), (64, 64, 3), (64, 64, 3)]


class _DiscriminatorFn(torch.autograd.Function):
    """logits = UNetDiscriminator(img).  Inputs: img, then conv_0.weight, conv_0.bias, the 8 weight_orig tensors,
    conv_9.weight, conv_9.bias (autograd leaves), then the 8 (u, v) buffer pairs (updated IN PLACE by the training-mode
    power iteration, like torch.nn.utils.spectral_norm's forward pre-hook)."""

    @staticmethod
    def forward(ctx, img, dtype, training, need_bwd, *tensors):
        from ._lib import DiscDesc
        lib = _lib.load()
        n, c, h, w = img.shape
        params, bufs = tensors[:12], tensors[12:]
        dev = img.device
        img32 = _f32c(img)
        p32 = [_f32c(p) for p in params]
        eff, sig, uv = [], [], []
        for k, (co, ci, ks) in enumerate(_DISC_SPECTRAL):
            worig = p32[2 + k]
            u, v = bufs[2 * k], bufs[2 * k + 1]
            weff = torch.empty_like(worig)
            sigma = torch.empty(1, dtype=torch.float32, device=dev)
            _lib.check(lib.vsr_spectral_norm(_ptr(worig), _ptr(u), _ptr(v), _ptr(weff), _ptr(sigma), co, ci * ks * ks, int(training),
                                             _stream()), "spectral_norm")
            eff.append(weff)
            sig.append(sigma)
            uv.append((u.detach().clone(), v.detach().clone()) if need_bwd else None)   # the constants of THIS forward's graph
        desc = DiscDesc(n, h, w, 64, dtype)
        nbytes = lib.vsr_disc_workspace_bytes(ctypes.byref(desc), int(need_bwd))
        if nbytes == 0:
            raise RuntimeError(f"vsrlab_amd: unsupported discriminator input for the HIP path: {(n, c, h, w)} "
                               "(3-channel frames, height and width multiples of 8, mid_ch = 64)")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        call = [p32[0], p32[1]] + eff + [p32[10], p32[11]]
        out = torch.empty((n, 1, h, w), dtype=torch.float32, device=dev)
        _lib.check(lib.vsr_disc_forward(ctypes.byref(desc), _ptr_array(call), 12, _ptr(img32), _ptr(out), _ptr(ws), ws.numel(),
                                        int(need_bwd), _stream()), "disc_forward")
        ctx.need_bwd = need_bwd
        if need_bwd:
            ctx.ws, ctx.img32, ctx.p32, ctx.sig, ctx.uv, ctx.desc_args = ws, img32, p32, sig, uv, (n, h, w, 64, dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        from ._lib import DiscDesc
        if not ctx.need_bwd:
            raise RuntimeError("vsrlab_amd: backward through a discriminator forward that ran without need_backward")
        if ctx.ws is None:
            raise RuntimeError("vsrlab_amd: trying to backward through the discriminator graph a second time")
        lib = _lib.load()
        desc = DiscDesc(*ctx.desc_args)
        p32 = ctx.p32
        need_p = [bool(ctx.needs_input_grad[4 + k]) for k in range(12)]
        any_p = any(need_p)
        geff = [torch.zeros_like(p) if any_p else None for p in p32]
        dimg = torch.empty_like(ctx.img32) if ctx.needs_input_grad[0] else None
        _lib.check(lib.vsr_disc_backward(ctypes.byref(desc), _ptr_array(geff), 12, _ptr(ctx.img32), _ptr(_f32c(dout)), _ptr(dimg),
                                         _ptr(ctx.ws), ctx.ws.numel(), _stream()), "disc_backward")
        grads = [None] * 12
        if any_p:
            grads[0], grads[1], grads[10], grads[11] = geff[0], geff[1], geff[10], geff[11]
            sn_scratch = torch.empty(1024, dtype=torch.float32, device=dout.device)      # VSR_SN_SCRATCH_FLOATS
            for k, (co, ci, ks) in enumerate(_DISC_SPECTRAL):
                gorig = torch.zeros_like(p32[2 + k])
                u, v = ctx.uv[k]
                _lib.check(lib.vsr_spectral_norm_backward(_ptr(geff[2 + k]), _ptr(p32[2 + k]), _ptr(u), _ptr(v), _ptr(ctx.sig[k]),
                                                          _ptr(gorig), co, ci * ks * ks, _ptr(sn_scratch), _stream()), "spectral_norm_backward")
                grads[2 + k] = gorig
        ctx.ws = None
        return (dimg, None, None, None) + tuple(g if need_p[k] else None for k, g in enumerate(grads)) + (None,) * 16


class _SpectralNormFn(torch.autograd.Function):
    """weight = weight_orig / sigma(weight_orig; u, v): torch.nn.utils.spectral_norm's ``compute_weight`` (one power iteration per
    training-mode forward, u and v updated IN PLACE, sigma = u . (W v) differentiated through W only) as ``vsr_spectral_norm`` /
    ``vsr_spectral_norm_backward`` -- what the discriminator function above does for its eight inner layers, for ONE layer."""

    @staticmethod
    def forward(ctx, worig, u, v, training):
        lib = _lib.load()
        w32 = _f32c(worig)
        co = w32.shape[0]
        k = w32.numel() // co
        weff = torch.empty_like(w32)
        sigma = torch.empty(1, dtype=torch.float32, device=w32.device)
        _lib.check(lib.vsr_spectral_norm(_ptr(w32), _ptr(u), _ptr(v), _ptr(weff), _ptr(sigma), co, k, int(training), _stream()), "spectral_norm")
        ctx.save_for_backward(w32, u.detach().clone(), v.detach().clone(), sigma)          # the constants of THIS forward's graph
        return weff

    @staticmethod
    def backward(ctx, geff):
        w32, u, v, sigma = ctx.saved_tensors
        lib = _lib.load()
        co = w32.shape[0]
        gorig = torch.zeros_like(w32)
        scratch = torch.empty(1024, dtype=torch.float32, device=w32.device)                 # VSR_SN_SCRATCH_FLOATS
        _lib.check(lib.vsr_spectral_norm_backward(_ptr(_f32c(geff)), _ptr(w32), _ptr(u), _ptr(v), _ptr(sigma), _ptr(gorig), co, w32.numel() // co,
                                                  _ptr(scratch), _stream()), "spectral_norm_backward")
        return gorig, None, None, None


def spectral_conv_forward(x: torch.Tensor, weight_orig: torch.Tensor, u: torch.Tensor, v: torch.Tensor, training: bool,
                          compute_dtype: Optional[str] = None) -> torch.Tensor:
    """SpectralConv.forward (core/modules/conv.py:6-13) on its own, for the shape the per-layer kernels serve: 3x3, stride 1,
    padding 1, 64 -> 64 (the discriminator's conv_7 / conv_8 shape; the wide and the 4x4 stride-2 layers exist only inside the
    discriminator engine).  Differentiable w.r.t. x and weight_orig; u / v are updated in place when ``training``."""
    _require_gpu(x, weight_orig)
    if tuple(weight_orig.shape) != (64, 64, 3, 3) or x.shape[1] != 64:
        raise NotImplementedError("standalone HIP SpectralConv: Conv2d(64, 64, 3, 1, 1); other shapes run inside UNetDiscriminator")
    weff = _SpectralNormFn.apply(weight_orig, u, v, bool(training))
    return _conv_layer_autograd(x, None, weff, None, 0, compute_dtype=compute_dtype)


def discriminator_forward(img: torch.Tensor, params: Sequence[torch.Tensor], buffers: Sequence[torch.Tensor], training: bool,
                          compute_dtype: Optional[str] = None) -> torch.Tensor:
    """``params``: conv_0.weight, conv_0.bias, conv_1..8 ``weight_orig``, conv_9.weight, conv_9.bias; ``buffers``:
    (weight_u, weight_v) of conv_1..8, updated in place when ``training``."""
    _require_gpu(img)
    if img.dim() != 4 or img.shape[1] != 3:
        raise ValueError("discriminator input must be (N,3,H,W)")
    if len(params) != 12 or len(buffers) != 16:
        raise ValueError("expected 12 parameter tensors and 16 spectral-norm buffers")
    need_bwd = torch.is_grad_enabled() and (img.requires_grad or any(p.requires_grad for p in params))
    return _DiscriminatorFn.apply(img, resolve_dtype(compute_dtype), bool(training), need_bwd, *params, *buffers)


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target):
        lib = _lib.load()
        x32 = _f32c(x)
        dx = torch.empty_like(x32)
        loss = torch.zeros((), dtype=torch.float32, device=x.device)
        _lib.check(lib.vsr_bce_with_logits(_ptr(x32), float(target), _ptr(dx), _ptr(loss), x32.numel(), _stream()), "bce_with_logits")
        ctx.save_for_backward(dx)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None


def bce_with_logits_const(x: torch.Tensor, target: float) -> torch.Tensor:
    """mean BCE-with-logits of ``x`` against the constant ``target`` (AdversarialLoss, core/losses.py:70-73)."""
    _require_gpu(x)
    return _BCEFn.apply(x, float(target))


# --------------------------------------------------------------------------------------------- #
# VRT window attention core (reference: vsr/models/VRT/modules/window_attention.py:116-162)
# --------------------------------------------------------------------------------------------- #
_MASK_BITS = {}


def _packed_mask(mask: torch.Tensor):
    """(m32, bits, value) for an attention mask: its contiguous fp32 form, and, for a {0, value} mask (compute_mask's 0 / -100)
    whose row length is a multiple of 32, the bit-packed form + value (else bits = None).
    Cached on the CALLER's tensor (identity, shape, dtype, version): ``TMSAG`` passes the lru-cached ``compute_mask`` tensor
    itself, so every shifted block of every forward is a hit; keying on the fp32 copy (round 2) missed whenever the caller's mask
    was not already contiguous fp32 and paid two host syncs + a re-pack per attention call.  A miss costs ONE host sync (the
    mask value travels in the launch descriptor); at most 16 masks are kept."""
    key = (mask.data_ptr(), tuple(mask.shape), mask.dtype, mask._version)
    hit = _MASK_BITS.get(key)
    if hit is None:
        m32 = _f32c(mask)
        nW, Nm, _ = m32.shape
        lo = m32.min()
        ok_t = ((m32 == 0) | (m32 == lo)).all() & (lo != 0)
        value, ok = (float(v) for v in torch.stack([lo, ok_t.to(lo.dtype)]).tolist())      # one transfer for both scalars
        bits = None
        if ok and Nm % 32 == 0:
            bits = torch.empty((nW, Nm, Nm // 32), dtype=torch.int32, device=m32.device)
            _lib.check(_lib.load().vsr_mask_pack(_ptr(m32), _ptr(bits), nW, Nm, _stream()), "mask_pack")
        else:
            value = 0.0
        if len(_MASK_BITS) >= 16:
            _MASK_BITS.clear()
        hit = (m32, bits, value, mask)                         # keeps `mask` alive: the key's data_ptr stays unique
        _MASK_BITS[key] = hit
    return hit[0], hit[1], hit[2]


class _WindowAttentionFn(torch.autograd.Function):
    """x_out = [mutual attention | self attention] of WindowAttention.forward, given the outputs of the qkv Linears.

    qkv_self (B_, N, 3C) ; qkv_mut (B_, N, 3C) or None ; table: relative_position_bias_table ((2Wd-1)(2Wh-1)(2Ww-1), heads);
    index: relative_position_index (int64) ; mask (nW, Nm, Nm) or None.  Returns (B_, N, 2C) with mutual attention,
    (B_, N, C) without -- the tensor the reference builds with torch.cat (:131-134) before ``proj``."""

    @staticmethod
    def forward(ctx, qkv_self, qkv_mut, table, index, mask, heads, scale, dtype):
        from ._lib import AttnDesc
        lib = _lib.load()
        tdt = _TORCH_DT[dtype]
        B, N, C3 = qkv_self.shape
        C = C3 // 3
        hd = C // heads
        mut = qkv_mut is not None
        dev = qkv_self.device
        qs = qkv_self.detach().to(tdt).contiguous()
        qm = qkv_mut.detach().to(tdt).contiguous() if mut else None
        Cout = 2 * C if mut else C
        out = torch.empty((B, N, Cout), dtype=tdt, device=dev)
        idx = index.contiguous()
        dense = torch.empty((heads, N, N), dtype=torch.float32, device=dev)
        _lib.check(lib.vsr_rpb_gather(_ptr(_f32c(table)), _ptr(idx), idx.shape[1], _ptr(dense), heads, N, _stream()), "rpb_gather")
        m32, bits, mval = _packed_mask(mask) if mask is not None else (None, None, 0.0)
        nW, Nm = (m32.shape[0], m32.shape[1]) if m32 is not None else (1, 0)
        if m32 is not None and (B % nW != 0 or Nm < N):
            raise ValueError("attention mask does not match the windows")
        calls = [(qs, 0, 0, 0, N, N, C if mut else 0, dense)]
        if mut:
            h2 = N // 2
            calls += [(qm, h2, 0, 0, h2, h2, 0, None),        # x1_aligned = attention(q2, k1, v1) -> tokens [0, N/2)
                      (qm, 0, h2, h2, h2, h2, 0, None)]       # x2_aligned = attention(q1, k2, v2) -> tokens [N/2, N)
        lses = []
        for (t, q0, k0, o0, nq, nk, coff, bias) in calls:
            packed = bits is not None and nq == nk and nq in (64, 128)
            d = AttnDesc(B, N, heads, hd, q0, k0, o0, nq, nk, Cout, coff, nW, Nm, float(scale), dtype, int(packed), mval)
            lse = torch.empty((B, heads, nq), dtype=torch.float32, device=dev)
            _lib.check(lib.vsr_window_attention_fwd(ctypes.byref(d), _ptr(t), _ptr(bias), _ptr(bits if packed else m32), _ptr(out), _ptr(lse),
                                                    _stream()), "window_attention_fwd")
            lses.append(lse)
        ctx.mask_bits, ctx.mask_value = bits, mval
        ctx.save_for_backward(qs, qm, dense, m32, idx, *lses)
        ctx.meta = (B, N, C, heads, hd, mut, Cout, nW, Nm, float(scale), dtype, tuple(table.shape), qkv_self.dtype)
        ctx.calls = [(c[1], c[2], c[3], c[4], c[5], c[6], c[7] is not None) for c in calls]
        return out.to(qkv_self.dtype)

    @staticmethod
    def backward(ctx, dout):
        from ._lib import AttnDesc
        lib = _lib.load()
        qs, qm, dense, m32, idx, *lses = ctx.saved_tensors
        B, N, C, heads, hd, mut, Cout, nW, Nm, scale, dtype, tshape, in_dt = ctx.meta
        tdt = _TORCH_DT[dtype]
        dev = dout.device
        do = dout.detach().to(tdt).contiguous()
        dqs = torch.empty_like(qs)
        dqm = torch.empty_like(qm) if mut else None
        ddense = torch.zeros_like(dense)
        for (q0, k0, o0, nq, nk, coff, has_bias), lse, (src, dst) in zip(ctx.calls, lses, [(qs, dqs)] + [(qm, dqm)] * 2):
            packed = ctx.mask_bits is not None and nq == nk and nq in (64, 128)
            d = AttnDesc(B, N, heads, hd, q0, k0, o0, nq, nk, Cout, coff, nW, Nm, scale, dtype, int(packed), ctx.mask_value)
            delta = torch.empty((B, heads, nq), dtype=torch.float32, device=dev)
            _lib.check(lib.vsr_window_attention_bwd(ctypes.byref(d), _ptr(src), _ptr(dense if has_bias else None),
                                                    _ptr(ctx.mask_bits if packed else m32), _ptr(do),
                                                    _ptr(lse), _ptr(delta), _ptr(dst), _ptr(ddense if has_bias else None), _stream()),
                       "window_attention_bwd")
        dtable = None
        if ctx.needs_input_grad[2]:
            dtable = torch.zeros(tshape, dtype=torch.float32, device=dev)
            _lib.check(lib.vsr_rpb_scatter(_ptr(ddense), _ptr(idx), idx.shape[1], _ptr(dtable), heads, N, _stream()), "rpb_scatter")
        return dqs.to(in_dt), (dqm.to(in_dt) if mut else None), dtable, None, None, None, None, None


def window_attention_core(qkv_self: torch.Tensor, qkv_mut: Optional[torch.Tensor], table: torch.Tensor, index: torch.Tensor,
                          mask: Optional[torch.Tensor], heads: int, scale: float, compute_dtype: Optional[str] = None) -> torch.Tensor:
    _require_gpu(qkv_self)
    B, N, C3 = qkv_self.shape
    if C3 % (3 * heads) != 0:
        raise ValueError("qkv width must be 3 * heads * head_dim")
    if (C3 // 3 // heads) > 32 or N % 32 != 0 or N > 384 or (qkv_mut is not None and N % 64 != 0):
        raise NotImplementedError("HIP window attention: head_dim <= 32 and windows of 64..384 tokens (VRT's (2,8,8) / (6,8,8))")
    return _WindowAttentionFn.apply(qkv_self, qkv_mut, table, index[:N, :N], mask, heads, scale, resolve_dtype(compute_dtype))


# --------------------------------------------------------------------------------------------- #
# single conv layers of the reference's building blocks (ConvReLU, the ResidualBlock stem, PixelShufflePack, SpynetModule's
# layers): forward vsr_conv_layer_fwd, backward vsr_conv_layer_bwd -- ordinary autograd modules, like the reference's
# --------------------------------------------------------------------------------------------- #
def conv_layer(x_pm: Optional[torch.Tensor], weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0, slope: float = 0.1,
               lr_planar: Optional[torch.Tensor] = None, pixel_shuffle: bool = False, planar_out: bool = False,
               dtype: Optional[int] = None, hw: Optional[Tuple[int, int, int]] = None) -> torch.Tensor:
    """``vsr_conv_layer_fwd``: one conv layer on a blocked pixel-major tensor (or the planar LR frames).  Returns a
    pixel-major tensor, or (N,cout,H,W) fp32 when ``planar_out``."""
    lib = _lib.load()
    cout, cin, ks, _ = weight.shape
    if x_pm is not None:
        n, h, w, cin_pm = _pm_dims(x_pm)
        dt = DT_BF16 if x_pm.dtype == torch.bfloat16 else DT_F32
        dev = x_pm.device
    else:
        n, h, w = hw
        cin_pm, dt, dev = 0, dtype, lr_planar.device
    wpack = torch.empty(49 * 64 * 64 * 4, dtype=_TORCH_DT[dt], device=dev)
    y_pm = y_pl = None
    cd = 0
    if planar_out:
        y_pl = torch.empty((n, cout, h, w), dtype=torch.float32, device=dev)
    else:
        cd = 64 if pixel_shuffle else max(16, cout)
        s = 2 if pixel_shuffle else 1
        y_pm = torch.empty((n, s * h, (s * w + 31) // 32, cd // 8, 32, 8), dtype=_TORCH_DT[dt], device=dev)
        y_pm.pm_w = s * w
    _lib.check(lib.vsr_conv_layer_fwd(dt, ks, _ptr(x_pm), cin_pm, _ptr(None if lr_planar is None else _f32c(lr_planar)), _ptr(_f32c(weight)),
                                      _ptr(None if bias is None else _f32c(bias)), cin, cout, _ptr(wpack), _ptr(y_pm), cd, _ptr(y_pl),
                                      act, float(slope), int(pixel_shuffle), n, h, w, _stream()), "conv_layer_fwd")
    return y_pl if planar_out else y_pm


class _ConvLayerFn(torch.autograd.Function):
    """y = act(conv(x [, lr]) + b) [+ PixelShuffle(2)] on planar (N,C,H,W) tensors; x / lr may be None (stems).  Backward =
    ``vsr_conv_layer_bwd``: data gradient(s), weight gradient, bias gradient."""

    @staticmethod
    def forward(ctx, x, lr, weight, bias, act, slope, pixel_shuffle, planar_out, dt):
        cout, cin = weight.shape[:2]
        ref = x if x is not None else lr
        n, _, h, w = ref.shape
        x_pm = None
        if x is not None:
            x_pm = to_pixel_major(x, dt, ((x.shape[1] + 15) // 16) * 16)
        lr32 = None if lr is None else _f32c(lr)
        y = conv_layer(x_pm, weight, bias, act=act, slope=slope, lr_planar=lr32, pixel_shuffle=pixel_shuffle, planar_out=planar_out,
                       dtype=dt, hw=(n, h, w))
        ctx.save_for_backward(x_pm, lr32, weight, y)
        ctx.meta = (act, float(slope), bool(pixel_shuffle), bool(planar_out), dt, n, h, w, cin, cout, None if x is None else x.shape[1],
                    x_pm.pm_w if x_pm is not None else None, bias is not None)
        if planar_out:
            return y
        return from_pixel_major(y, 64 if pixel_shuffle else cout)

    @staticmethod
    def backward(ctx, dout):
        x_pm, lr32, weight, y = ctx.saved_tensors
        act, slope, ps, planar, dt, n, h, w, cin, cout, xc, xw, has_bias = ctx.meta
        lib = _lib.load()
        dev = dout.device
        if x_pm is not None:
            x_pm.pm_w = xw
        need_x, need_lr, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        cd = 0 if planar else (64 if ps else max(16, cout))
        dy_pm = dy_pl = y_pm = y_pl = None
        if planar:
            dy_pl, y_pl = _f32c(dout), y
        else:
            dy_pm = to_pixel_major(dout, dt, cd)
            y_pm = y
        dx_pm = None
        if need_x and x_pm is not None:
            dx_pm = torch.empty_like(x_pm)
            dx_pm.pm_w = xw
        dlr = torch.empty_like(lr32) if (need_lr and lr32 is not None) else None
        gw = torch.empty(weight.shape, dtype=torch.float32, device=dev) if (need_w or need_b) else None
        gb = torch.empty(cout, dtype=torch.float32, device=dev) if (need_b and has_bias) else None
        nbytes = lib.vsr_conv_layer_bwd_scratch_bytes(dt, n, h, w, int(ps))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ks = weight.shape[2]
        _lib.check(lib.vsr_conv_layer_bwd(dt, ks, _ptr(x_pm), 0 if x_pm is None else x_pm.shape[3] * 8, _ptr(lr32), _ptr(_f32c(weight)), cin, cout,
                                          _ptr(y_pm), _ptr(y_pl), _ptr(dy_pm), _ptr(dy_pl), cd, act, slope, int(ps), _ptr(dx_pm), _ptr(dlr),
                                          _ptr(gw), _ptr(gb), _ptr(scratch), nbytes, n, h, w, _stream()), "conv_layer_bwd")
        dx = from_pixel_major(dx_pm, xc) if dx_pm is not None else None
        return dx, dlr, (gw if need_w else None), gb, None, None, None, None, None


def _conv_layer_autograd(x, lr, weight, bias, act, slope=0.1, pixel_shuffle=False, planar_out=False, compute_dtype=None):
    return _ConvLayerFn.apply(x, lr, weight, bias, act, slope, pixel_shuffle, planar_out, resolve_dtype(compute_dtype))


def conv_relu_forward(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], compute_dtype: Optional[str] = None) -> torch.Tensor:
    """ConvReLU.forward (core/modules/conv.py:15-22) for the SPyNet layer shapes (7x7) and 64 -> 64 (3x3 / 1x1); differentiable."""
    _require_gpu(x)
    cout = weight.shape[0]
    return _conv_layer_autograd(x, None, weight, bias, 1, planar_out=cout <= 4, compute_dtype=compute_dtype)


def spynet_module_forward(x: torch.Tensor, params: Sequence[torch.Tensor], last_relu: bool = True,
                          compute_dtype: Optional[str] = None) -> torch.Tensor:
    """SpynetModule.forward (spynet.py:13-21): (N,8,h,w) -> (N,2,h,w) through the five 7x7 layers; differentiable w.r.t. x and
    the ten parameter tensors."""
    _require_gpu(x)
    if x.shape[1] != 8 or len(params) != 10:
        raise ValueError("SpynetModule takes (N,8,h,w) and has 10 parameter tensors")
    y = x
    for j in range(4):
        y = _conv_layer_autograd(y, None, params[2 * j], params[2 * j + 1], 1, compute_dtype=compute_dtype)
    return _conv_layer_autograd(y, None, params[8], params[9], 1 if last_relu else 0, planar_out=True, compute_dtype=compute_dtype)


def pixel_shuffle_pack_forward(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], compute_dtype: Optional[str] = None):
    """PixelShufflePack.forward (upsampling.py:10-12): conv3x3 64 -> 256 + PixelShuffle(2); the shuffle is the store pattern;
    differentiable."""
    _require_gpu(x)
    if x.shape[1] != 64 or tuple(weight.shape) != (256, 64, 3, 3):
        raise NotImplementedError("the HIP PixelShufflePack is built for 64 -> 64 channels, scale 2 (the reference's use)")
    return _conv_layer_autograd(x, None, weight, bias, 0, pixel_shuffle=True, compute_dtype=compute_dtype)


def residual_block_forward(x: torch.Tensor, stem_w: torch.Tensor, stem_b: torch.Tensor, blocks: Sequence[Tuple[torch.Tensor, ...]],
                           compute_dtype: Optional[str] = None) -> torch.Tensor:
    """ResidualBlock.forward (core/modules/conv.py:94-103) on x = cat([lr(3), feat(64)]) (the trunks) or x = lr (the pre-clean
    stack): conv3x3 + LeakyReLU(0.1), then the ResidualConv blocks ((w1, b1, w2, b2) each); differentiable (the stem through
    vsr_conv_layer_bwd, the blocks through the ResidualConv function)."""
    _require_gpu(x)
    n, cin, h, w = x.shape
    if cin == 67:
        y = _conv_layer_autograd(x[:, 3:].contiguous(), x[:, :3].contiguous(), stem_w, stem_b, 2, compute_dtype=compute_dtype)
    elif cin == 3:
        y = _conv_layer_autograd(None, x, stem_w, stem_b, 2, compute_dtype=compute_dtype)
    else:
        raise NotImplementedError("the HIP ResidualBlock stem takes 3 (pre-clean) or 3 + 64 (trunk) input channels")
    for (w1, b1, w2, b2) in blocks:
        y = residual_conv(y, w1, b1, w2, b2, compute_dtype)
    return y

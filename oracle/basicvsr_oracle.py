"""CPU oracle (TEST INFRASTRUCTURE) for the BasicVSR forward/backward hot path.

A from-scratch, purely functional fp32 restatement of the reference algorithm
(santurini/vsrlab), written against ``torch.nn.functional`` on CPU.  It is the
checker for the HIP path and the ``cpu_baseline`` ("port") leg of ``bench.py``;
it is never imported by the product package ``vsrlab_amd``.

Parity pin: this file is checked (tests/test_oracle_golden.py) against golden
vectors in ``tests/golden/*.npz`` that were produced by importing the real
reference in the build container (``tests/golden/make_golden.py``).  The
reference itself ships no tests/fixtures, so those vectors are the only pin
(SURVEY.md section 8c).

All parameters are passed as a ``state_dict``-style mapping with the reference's
own key names (``backward_resblocks.conv.0.weight`` ...), so a reference
checkpoint can be fed in unchanged.

Reference citations are relative to /root/reference/src .
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Mapping, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

SPYNET_MEAN = (0.485, 0.456, 0.406)  # vsr/models/RealBasicVSR/modules/spynet.py:30
SPYNET_STD = (0.229, 0.224, 0.225)   # vsr/models/RealBasicVSR/modules/spynet.py:31
SPYNET_CHANNELS = ((8, 32), (32, 64), (64, 32), (32, 16), (16, 2))  # spynet.py:16-18
SPYNET_LEVELS = 6                    # spynet.py:29


# --------------------------------------------------------------------------- #
# optional emulation of the perf build's storage precision
# --------------------------------------------------------------------------- #
# The HIP perf build stores every activation and activation-gradient in bf16 and feeds the
# matrix cores bf16 weights, accumulating in fp32.  ``emulate_bf16()`` makes this oracle round at
# exactly those points (same algorithm, fp32 arithmetic, bf16 storage), which gives the bf16 build a
# TIGHT checker instead of a loose "bf16 is noisy" tolerance.  Default: no rounding (the reference).
class _RoundBoth(torch.autograd.Function):          # stored activation: value and its gradient are bf16
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundGrad(torch.autograd.Function):          # fp32 tensor whose cotangent is staged as bf16
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


_EMULATE_BF16 = False


class emulate_bf16:
    def __enter__(self):
        global _EMULATE_BF16
        self._prev = _EMULATE_BF16
        _EMULATE_BF16 = True

    def __exit__(self, *exc):
        global _EMULATE_BF16
        _EMULATE_BF16 = self._prev


def _q(x: Tensor) -> Tensor:        # stored activation
    return _RoundBoth.apply(x) if _EMULATE_BF16 else x


def _wq(w: Tensor) -> Tensor:       # weight as fed to the MFMA (its gradient stays fp32)
    return w + (w.to(torch.bfloat16).to(w.dtype) - w).detach() if _EMULATE_BF16 else w


def _iq(x: Tensor) -> Tensor:       # fp32 boundary tensor staged as bf16 (LR frame into the stem conv)
    return x.to(torch.bfloat16).to(x.dtype) if _EMULATE_BF16 else x


def _gq(x: Tensor) -> Tensor:       # fp32 output whose cotangent is staged as bf16 (SR frame)
    return _RoundGrad.apply(x) if _EMULATE_BF16 else x


# --------------------------------------------------------------------------- #
# deterministic, init-order-independent parameters
# --------------------------------------------------------------------------- #
def basicvsr_param_shapes(mid_channels: int = 64, res_blocks: int = 30,
                          upscale: int = 4) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape for the reference ``BasicVSR`` (basicvsr.py:12-23)."""
    c = mid_channels
    shapes: Dict[str, Tuple[int, ...]] = {}
    for trunk in ("backward_resblocks", "forward_resblocks"):
        shapes[f"{trunk}.conv.0.weight"] = (c, c + 3, 3, 3)      # core/modules/conv.py:97
        shapes[f"{trunk}.conv.0.bias"] = (c,)
        for i in range(res_blocks):                              # core/modules/conv.py:85-86
            for j in (1, 2):
                shapes[f"{trunk}.res_block.{i}.conv{j}.weight"] = (c, c, 3, 3)
                shapes[f"{trunk}.res_block.{i}.conv{j}.bias"] = (c,)
    shapes["point_conv.0.weight"] = (c, 2 * c, 1, 1)             # basicvsr.py:18
    shapes["point_conv.0.bias"] = (c,)
    for k in range(upscale // 2):                                # basicvsr.py:19
        shapes[f"upsample.{k}.upconv.weight"] = (4 * c, c, 3, 3)  # core/modules/upsampling.py:7
        shapes[f"upsample.{k}.upconv.bias"] = (4 * c,)
    shapes["conv_last.0.weight"] = (64, c, 3, 3)                 # basicvsr.py:20
    shapes["conv_last.0.bias"] = (64,)
    shapes["conv_last.2.weight"] = (3, 64, 3, 3)                 # basicvsr.py:21
    shapes["conv_last.2.bias"] = (3,)
    shapes.update({f"spynet.{k}": v for k, v in spynet_param_shapes().items()})
    return shapes


def spynet_param_shapes() -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape for the reference ``Spynet`` (spynet.py:24-31)."""
    shapes: Dict[str, Tuple[int, ...]] = {}
    for lvl in range(SPYNET_LEVELS):
        for j, (ci, co) in enumerate(SPYNET_CHANNELS):
            shapes[f"basic_module.{lvl}.basic_module.{j}.conv.0.weight"] = (co, ci, 7, 7)
            shapes[f"basic_module.{lvl}.basic_module.{j}.conv.0.bias"] = (co,)
    shapes["mean"] = (1, 3, 1, 1)
    shapes["std"] = (1, 3, 1, 1)
    return shapes


def keyed_tensor(key: str, shape: Tuple[int, ...]) -> Tensor:
    """Value of parameter ``key``: depends only on the key's name and shape."""
    if key.endswith("mean"):
        return torch.tensor(SPYNET_MEAN, dtype=torch.float32).view(shape)
    if key.endswith("std"):
        return torch.tensor(SPYNET_STD, dtype=torch.float32).view(shape)
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
    v = torch.randn(shape, generator=g, dtype=torch.float32)
    if key.endswith("bias"):
        return v * 0.01
    if key.endswith(("weight_u", "weight_v")):          # spectral_norm's power-iteration vectors: unit norm
        return v / v.norm()
    if key.endswith("relative_position_bias_table"):    # VRT window attention: O(1) biases so that they matter
        return v * 0.5
    if len(shape) == 1:                                 # LayerNorm weight
        return 1.0 + 0.1 * v
    if len(shape) == 2:                                 # nn.Linear weight (out, in)
        return v * (1.0 / math.sqrt(shape[1]))
    fan_in = shape[1] * shape[2] * shape[3]
    # 0.7/sqrt(fan_in) keeps a 30-block trunk O(1); the SPyNet convs get a larger gain so
    # that the synthetic flows reach a few pixels and the warps are genuinely exercised
    gain = 1.5 if "basic_module" in key else 0.7
    return v * (gain / math.sqrt(fan_in))


def keyed_state_dict(shapes: Mapping[str, Tuple[int, ...]]) -> Dict[str, Tensor]:
    return {k: keyed_tensor(k, tuple(s)) for k, s in shapes.items()}


# --------------------------------------------------------------------------- #
# flow_warp  (vsr/models/RealBasicVSR/modules/spynet.py:95-106)
# --------------------------------------------------------------------------- #
def flow_warp(x: Tensor, flow: Tensor, padding_mode: str = "zeros") -> Tensor:
    """out[n,c,y,x] = bilinear(x[n,c], (x + flow[n,0,y,x], y + flow[n,1,y,x])).

    ``flow`` is planar ``(N,2,H,W)`` with channel 0 = horizontal displacement (the
    reference passes the same data permuted to ``(N,H,W,2)``, basicvsr.py:54).
    The reference builds a mesh grid, normalises by ``max(dim-1,1)`` and calls
    ``grid_sample(align_corners=True)`` (spynet.py:97-105); with align_corners the
    normalisation round-trips to pixel units, so the sample position is exactly
    ``(x+fx, y+fy)``.  ``zeros``: every out-of-image *tap* contributes 0.
    ``border``: the sample coordinate is clamped to ``[0, dim-1]`` first.
    """
    n, c, h, w = x.shape
    ys, xs = torch.meshgrid(torch.arange(h, dtype=x.dtype), torch.arange(w, dtype=x.dtype),
                            indexing="ij")
    # reproduce the reference's normalise/un-normalise arithmetic so rounding agrees
    gx = 2.0 * (xs + flow[:, 0]) / max(w - 1, 1) - 1.0
    gy = 2.0 * (ys + flow[:, 1]) / max(h - 1, 1) - 1.0
    px = (gx + 1.0) * 0.5 * (w - 1)
    py = (gy + 1.0) * 0.5 * (h - 1)
    if padding_mode == "border":
        px = px.clamp(0, w - 1)
        py = py.clamp(0, h - 1)
    elif padding_mode != "zeros":
        raise ValueError(padding_mode)
    x0 = torch.floor(px)
    y0 = torch.floor(py)
    wx1 = px - x0
    wy1 = py - y0
    wx0 = 1.0 - wx1
    wy0 = 1.0 - wy1
    flat = x.reshape(n, c, h * w)
    out = torch.zeros_like(x)
    for dy, wy in ((0, wy0), (1, wy1)):
        for dx, wx in ((0, wx0), (1, wx1)):
            xi = x0 + dx
            yi = y0 + dy
            valid = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1)
            idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).long()
            tap = torch.gather(flat, 2, idx.view(n, 1, h * w).expand(n, c, h * w)).view(n, c, h, w)
            out = out + tap * (wx * wy * valid.to(x.dtype)).unsqueeze(1)
    return out


# --------------------------------------------------------------------------- #
# SPyNet  (vsr/models/RealBasicVSR/modules/spynet.py:13-93)
# --------------------------------------------------------------------------- #
def _spynet_level(sd: Mapping[str, Tensor], prefix: str, lvl: int, x: Tensor) -> Tensor:
    """5 x (conv7x7 pad 3 + ReLU); the ReLU also follows the LAST conv (spynet.py:16-18)."""
    for j in range(len(SPYNET_CHANNELS)):
        k = f"{prefix}basic_module.{lvl}.basic_module.{j}.conv.0."
        x = F.relu(F.conv2d(x, _wq(sd[k + "weight"]), sd[k + "bias"], stride=1, padding=3))
        if j < len(SPYNET_CHANNELS) - 1:
            x = _q(x)          # the last layer's 2 channels stay fp32 (planar flow)
    return x


def spynet_compute_flow(sd: Mapping[str, Tensor], ref: Tensor, supp: Tensor,
                        prefix: str = "") -> Tensor:
    """Coarse-to-fine flow on /32-aligned frames (spynet.py:38-67)."""
    mean = sd[prefix + "mean"]
    std = sd[prefix + "std"]
    n, _, h, w = ref.shape
    refs = [(ref - mean) / std]
    supps = [(supp - mean) / std]
    for _ in range(SPYNET_LEVELS - 1):                             # spynet.py:42-45
        refs.append(F.avg_pool2d(refs[-1], 2, 2))
        supps.append(F.avg_pool2d(supps[-1], 2, 2))
    refs.reverse()
    supps.reverse()
    flow = ref.new_zeros(n, 2, h // 32, w // 32)                   # spynet.py:49
    for lvl in range(SPYNET_LEVELS):
        if lvl == 0:
            flow_up = flow
        else:                                                      # spynet.py:54
            flow_up = F.interpolate(flow, scale_factor=2, mode="bilinear", align_corners=True) * 2.0
        warped = flow_warp(supps[lvl], flow_up, padding_mode="border")  # spynet.py:58-60
        res = _spynet_level(sd, prefix, lvl, _q(torch.cat([refs[lvl], warped, flow_up], 1)))
        flow = flow_up + res                                       # spynet.py:65
    return flow


def spynet_forward(sd: Mapping[str, Tensor], ref: Tensor, supp: Tensor, prefix: str = "") -> Tensor:
    """Flow (N,2,h,w) from ``ref`` to ``supp`` at the input resolution (spynet.py:69-93)."""
    h, w = ref.shape[2:]
    w_up = w if w % 32 == 0 else 32 * (w // 32 + 1)
    h_up = h if h % 32 == 0 else 32 * (h // 32 + 1)
    ref_u = F.interpolate(ref, size=(h_up, w_up), mode="bilinear", align_corners=False)
    supp_u = F.interpolate(supp, size=(h_up, w_up), mode="bilinear", align_corners=False)
    flow = F.interpolate(spynet_compute_flow(sd, ref_u, supp_u, prefix), size=(h, w),
                         mode="bilinear", align_corners=False)
    scale = torch.tensor([float(w) / float(w_up), float(h) / float(h_up)],
                         dtype=flow.dtype).view(1, 2, 1, 1)        # spynet.py:90-91
    return flow * scale


# --------------------------------------------------------------------------- #
# trunk / reconstruction blocks
# --------------------------------------------------------------------------- #
def residual_conv(sd: Mapping[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """x + conv2(relu(conv1(x)))  (core/modules/conv.py:82-92)."""
    y = _q(F.relu(F.conv2d(x, _wq(sd[prefix + "conv1.weight"]), sd[prefix + "conv1.bias"], padding=1)))
    y = F.conv2d(y, _wq(sd[prefix + "conv2.weight"]), sd[prefix + "conv2.bias"], padding=1)
    return _q(x + y)


def residual_block(sd: Mapping[str, Tensor], prefix: str, x: Tensor, blocks: int) -> Tensor:
    """conv3x3 + LeakyReLU(0.1), then ``blocks`` ResidualConv (core/modules/conv.py:94-103)."""
    x = _q(F.leaky_relu(F.conv2d(x, _wq(sd[prefix + "conv.0.weight"]), sd[prefix + "conv.0.bias"], padding=1), 0.1))
    for i in range(blocks):
        x = residual_conv(sd, f"{prefix}res_block.{i}.", x)
    return x


def pixel_shuffle_pack(sd: Mapping[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """conv3x3 C->4C then PixelShuffle(2), no activation (core/modules/upsampling.py:4-12)."""
    y = F.conv2d(x, _wq(sd[prefix + "upconv.weight"]), sd[prefix + "upconv.bias"], padding=1)
    return _q(F.pixel_shuffle(y, 2))


def count_res_blocks(sd: Mapping[str, Tensor], prefix: str = "") -> int:
    n = 0
    while f"{prefix}forward_resblocks.res_block.{n}.conv1.weight" in sd:
        n += 1
    return n


def count_upsample(sd: Mapping[str, Tensor], prefix: str = "") -> int:
    n = 0
    while f"{prefix}upsample.{n}.upconv.weight" in sd:
        n += 1
    return n


# --------------------------------------------------------------------------- #
# BasicVSR  (vsr/models/RealBasicVSR/modules/basicvsr.py:30-83)
# --------------------------------------------------------------------------- #
def basicvsr_compute_flow(sd: Mapping[str, Tensor], lrs: Tensor, prefix: str = "") -> Tuple[Tensor, Tensor]:
    """(flows_forward, flows_backward), each (n,t-1,2,h,w)  (basicvsr.py:30-44)."""
    n, t, c, h, w = lrs.shape
    a = lrs[:, :-1].reshape(-1, c, h, w)
    b = lrs[:, 1:].reshape(-1, c, h, w)
    sp = prefix + "spynet."
    flow_backward = spynet_forward(sd, a, b, sp)    # ref = frame i, supp = frame i+1
    flow_forward = spynet_forward(sd, b, a, sp)     # ref = frame i+1, supp = frame i
    return flow_forward.view(n, t - 1, 2, h, w), flow_backward.view(n, t - 1, 2, h, w)


def basicvsr_forward(sd: Mapping[str, Tensor], lrs: Tensor, prefix: str = "",
                     flows: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """``BasicVSR.forward`` (basicvsr.py:39-83): (n,t,3,h,w) -> (n,t,3,s*h,s*w)."""
    n, t, _, h, w = lrs.shape
    rb = count_res_blocks(sd, prefix)
    n_up = count_upsample(sd, prefix)
    mid = sd[prefix + "point_conv.0.weight"].shape[0]
    if flows is None:
        flows = basicvsr_compute_flow(sd, lrs, prefix)
    flows_forward, flows_backward = flows

    feats_bwd = [None] * t
    feat = lrs.new_zeros(n, mid, h, w)
    for i in range(t - 1, -1, -1):                                  # basicvsr.py:48-60
        if i < t - 1:
            feat = _q(flow_warp(feat, flows_backward[:, i]))
        feat = residual_block(sd, prefix + "backward_resblocks.", torch.cat([_iq(lrs[:, i]), feat], 1), rb)
        feats_bwd[i] = feat

    outs = []
    feat = torch.zeros_like(feat)
    for i in range(t):                                              # basicvsr.py:64-82
        if i > 0:
            feat = _q(flow_warp(feat, flows_forward[:, i - 1]))
        feat = residual_block(sd, prefix + "forward_resblocks.", torch.cat([_iq(lrs[:, i]), feat], 1), rb)
        o = F.conv2d(torch.cat([feats_bwd[i], feat], 1), _wq(sd[prefix + "point_conv.0.weight"]),
                     sd[prefix + "point_conv.0.bias"])
        o = _q(F.leaky_relu(o, 0.1))
        for k in range(n_up):
            o = pixel_shuffle_pack(sd, f"{prefix}upsample.{k}.", o)
        o = _q(F.leaky_relu(F.conv2d(o, _wq(sd[prefix + "conv_last.0.weight"]), sd[prefix + "conv_last.0.bias"],
                                     padding=1), 0.1))
        o = _gq(F.conv2d(o, _wq(sd[prefix + "conv_last.2.weight"]), sd[prefix + "conv_last.2.bias"], padding=1))
        base = F.interpolate(lrs[:, i], scale_factor=2 ** n_up, mode="bilinear", align_corners=False)
        outs.append(o + base)
    return torch.stack(outs, 1)


def charbonnier(x: Tensor, y: Tensor, eps: float = 1e-9) -> Tensor:
    """mean(sqrt((x-y)^2 + eps))  (core/losses.py:10-18)."""
    d = x - y
    return torch.mean(torch.sqrt(d * d + eps))


# --------------------------------------------------------------------------- #
# RealBasicVSR pre-clean stack (vsr/models/RealBasicVSR/realbasicvsr.py:5-30)
# --------------------------------------------------------------------------- #
def cleaner_forward(sd: Mapping[str, Tensor], lr: Tensor, prefix: str = "cleaner.", steps: int = 3) -> Tensor:
    """IterativeRefinement: x <- x + conv(resblock(x)), ``steps`` times.

    Returns a fresh tensor; the reference updates its input in place
    (realbasicvsr.py:29) - the values are identical.
    """
    n, t, c, h, w = lr.shape
    blocks = 0
    while f"{prefix}resblock.res_block.{blocks}.conv1.weight" in sd:
        blocks += 1
    x = lr.reshape(-1, c, h, w)
    for _ in range(steps):
        # (bf16 emulation points of the HIP perf build: the fp32 planar frame is staged as bf16 into the stem, the 64 -> 3 conv
        # is fed bf16 weights and stages its fp32 cotangent as bf16; the running frame x itself stays fp32)
        r = residual_block(sd, prefix + "resblock.", _iq(x), blocks)
        x = x + _gq(F.conv2d(r, _wq(sd[prefix + "conv.weight"]), sd[prefix + "conv.bias"], padding=1))
    return x.view(n, t, c, h, w)


def realbasicvsr_forward(sd: Mapping[str, Tensor], lr: Tensor) -> Tuple[Tensor, Tensor]:
    """``sr, lq = RealBasicVSR(lr)``  (realbasicvsr.py:11-15)."""
    lq = cleaner_forward(sd, lr)
    return basicvsr_forward(sd, lq, prefix="basicvsr."), lq


def fwd_bwd(sd: Mapping[str, Tensor], lrs: Tensor, hr: Tensor, train_flow: bool = False,
            cot: Optional[Tensor] = None):
    """One forward + loss + backward; returns (sr, loss, grads by key).

    loss = Charbonnier(sr, hr) (the training loss, train.py:93-95) unless a cotangent
    ``cot`` is given, in which case the back-propagated scalar is mean(sr*cot) - a linear,
    well-conditioned functional used to pin gradients (Charbonnier is still returned).
    """
    leaves = {}
    for k, v in sd.items():
        v = v.detach().clone()
        trainable = v.is_floating_point() and not k.endswith(("mean", "std")) and \
            (train_flow or "spynet" not in k)                       # basicvsr.py:25-28
        v.requires_grad_(trainable)
        leaves[k] = v
    sr = basicvsr_forward(leaves, lrs)
    loss = charbonnier(sr, hr)
    (loss if cot is None else torch.mean(sr * cot)).backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    return sr.detach(), loss.detach(), grads


def cleaner_param_shapes(mid_channels: int = 64, blocks: int = 20, prefix: str = "cleaner.") -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape of ``IterativeRefinement`` (realbasicvsr.py:18-22)."""
    c = mid_channels
    shapes: Dict[str, Tuple[int, ...]] = {f"{prefix}resblock.conv.0.weight": (c, 3, 3, 3), f"{prefix}resblock.conv.0.bias": (c,)}
    for i in range(blocks):
        for j in (1, 2):
            shapes[f"{prefix}resblock.res_block.{i}.conv{j}.weight"] = (c, c, 3, 3)
            shapes[f"{prefix}resblock.res_block.{i}.conv{j}.bias"] = (c,)
    shapes[f"{prefix}conv.weight"] = (3, c, 3, 3)
    shapes[f"{prefix}conv.bias"] = (3,)
    return shapes

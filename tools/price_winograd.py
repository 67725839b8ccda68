#!/usr/bin/env python3
"""Pricing Winograd F(2x2, 3x3) for the trunk's 3x3 64 -> 64 convolutions (round-3 VERDICT next #1c), part (i): NUMERICS, on the CPU.

The bf16 build is judged by the noise-floor criterion of tests/test_hip_parity.py: its error against the fp64 oracle may not exceed
1.5 x the error of the oracle evaluated with bf16 STORAGE at the same points (oracle.emulate_bf16: bf16 weights and activations
into an fp32-accumulating direct convolution).  A Winograd kernel on the matrix cores would feed them bf16 TRANSFORMED operands:
U = bf16(G w G^T) (16 instead of 9 matrices per layer) and V = bf16(B^T d B) (sums of four bf16 activations, rounded again), fp32
accumulation over the 64 input channels, the output transform A^T M A in fp32.  This script evaluates BASELINE config 1
(n=2, t=5, 64x64, 30 blocks) three ways -- fp64 (truth), direct bf16 emulation, Winograd bf16 emulation of the 120 trunk layers --
and prints err(winograd) / err(direct) for sr and for the 254 gradients: the kernel could only be built if that ratio stays < 1.5.

    python tools/price_winograd.py
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import basicvsr_oracle as O  # noqa: E402  (a pricing tool: checker-side code only)

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def rt(x):          # round to bf16, straight-through gradient
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


def winograd_conv(x, w, b):
    """conv2d(x, w, b, padding=1) through F(2x2, 3x3) with bf16 transformed operands and fp32 accumulation (H, W even)."""
    n, c, h, wd = x.shape
    U = rt(torch.einsum("ai,ocij,bj->ocab", G, w, G))                                   # (O, C, 4, 4), bf16
    p = F.unfold(F.pad(x, (1, 1, 1, 1)), kernel_size=4, stride=2)                        # (n, c*16, T)
    T = p.shape[-1]
    d = p.view(n, c, 4, 4, T)
    V = rt(torch.einsum("ai,ncijt,bj->ncabt", BT, d, BT))                                # bf16 transformed input tiles
    M = torch.einsum("ocab,ncabt->noabt", U, V)                                          # fp32 accumulation over c
    Y = torch.einsum("ia,noabt,jb->noijt", AT, M, AT)                                    # (n, o, 2, 2, T), fp32
    th, tw = h // 2, wd // 2
    y = Y.view(n, -1, 2, 2, th, tw).permute(0, 1, 4, 2, 5, 3).reshape(n, -1, h, wd)
    return y + b.view(1, -1, 1, 1)


def residual_conv_winograd(sd, prefix, x):
    y = O._q(F.relu(winograd_conv(x, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"])))
    y = winograd_conv(y, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"])
    return O._q(x + y)


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    # self-check of the transform in fp32: equal to conv2d
    g = torch.Generator().manual_seed(3)
    x, w, b = torch.randn(1, 8, 12, 16, generator=g), torch.randn(8, 8, 3, 3, generator=g), torch.randn(8, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    rt_saved = globals()["rt"]
    globals()["rt"] = lambda t: t
    assert float((winograd_conv(x, w, b) - ref).abs().max()) < 1e-4
    globals()["rt"] = rt_saved

    rb, shape = 30, (2, 5, 3, 64, 64)
    sd = O.keyed_state_dict(O.basicvsr_param_shapes(64, rb, 4))
    g = torch.Generator().manual_seed(0)
    lrs = torch.rand(*shape, generator=g)
    hr = torch.rand(2, 5, 3, 256, 256, generator=g)
    cot = torch.rand(2, 5, 3, 256, 256, generator=g) * 2 - 1
    sr64, _, g64 = O.fwd_bwd({k: v.double() for k, v in sd.items()}, lrs.double(), hr.double(), cot=cot.double())
    with O.emulate_bf16():
        sr_d, _, g_d = O.fwd_bwd(sd, lrs, hr, cot=cot)
        direct = O.residual_conv
        O.residual_conv = residual_conv_winograd
        try:
            sr_w, _, g_w = O.fwd_bwd(sd, lrs, hr, cot=cot)
        finally:
            O.residual_conv = direct
    keys = sorted(g64)
    cat = lambda d: torch.cat([d[k].double().flatten() for k in keys])
    rel = lambda a, r: float((a.double() - r).norm() / r.norm())
    e_sr_d, e_sr_w = rel(sr_d, sr64), rel(sr_w, sr64)
    e_g_d, e_g_w = rel(cat(g_d), cat(g64)), rel(cat(g_w), cat(g64))
    worst = max((rel(g_w[k], g64[k]) / max(rel(g_d[k], g64[k]), 1e-3), k) for k in keys)
    print(f"config 1, 30 blocks: sr rel-L2 error   direct bf16 {e_sr_d:.3e}   winograd bf16 {e_sr_w:.3e}   ratio {e_sr_w / e_sr_d:.2f}")
    print(f"                     gradients (254 tensors, global rel-L2)   direct {e_g_d:.3e}   winograd {e_g_w:.3e}   ratio {e_g_w / e_g_d:.2f}")
    print(f"                     worst per-tensor ratio {worst[0]:.2f} ({worst[1]})    [criterion: global <= 1.5, per tensor <= 2.5]")


if __name__ == "__main__":
    main()

"""Micro-benchmark of the fused window attention at BASELINE config 5's stage shape (for rocprofv3 / PMC runs).
    python tools/bench_attention.py [iters]   -- 7360 windows x 128 tokens, dim 120, 6 heads (head_dim 20), bf16"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    import vsrlab_amd
    from vsrlab_amd import functional as VF
    from vsrlab_amd._lib import AttnDesc
    lib = vsrlab_amd._lib.load()
    dev = torch.device("cuda:0")
    B, N, heads, hd = 8 * 23 * 40, 128, 6, 20
    C = heads * hd
    qkv = torch.randn(B, N, 3, heads, hd, device=dev).to(torch.bfloat16)
    bias = torch.randn(heads, N, N, device=dev)
    nW = 920
    mask = torch.zeros(nW, N, N, device=dev)
    out = torch.empty(B, N, C, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, heads, N, device=dev)
    d = AttnDesc(B, N, heads, hd, 0, 0, 0, N, N, C, 0, nW, N, hd ** -0.5, VF.DT_BF16)
    st = VF._stream()
    bits = torch.empty(nW, N, N // 32, dtype=torch.int32, device=dev)
    lib.vsr_mask_pack(VF._ptr(mask), VF._ptr(bits), nW, N, st)
    dp = AttnDesc(B, N, heads, hd, 0, 0, 0, N, N, C, 0, nW, N, hd ** -0.5, VF.DT_BF16, 1, -100.0)
    variants = {"bias+mask": (bias, mask), "bias+bits": (bias, bits), "bias": (bias, None), "plain": (None, None)}
    for name, (bb, mm) in variants.items():
        d = dp if name == "bias+bits" else AttnDesc(B, N, heads, hd, 0, 0, 0, N, N, C, 0, nW, N, hd ** -0.5, VF.DT_BF16)
        for _ in range(2):
            lib.vsr_window_attention_fwd(ctypes.byref(d), VF._ptr(qkv), VF._ptr(bb), VF._ptr(mm), VF._ptr(out), VF._ptr(lse), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            lib.vsr_window_attention_fwd(ctypes.byref(d), VF._ptr(qkv), VF._ptr(bb), VF._ptr(mm), VF._ptr(out), VF._ptr(lse), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        gb = (qkv.numel() * 2 + out.numel() * 2) / 1e9
        print(f"fwd {name:10s}: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s of qkv+out  {4.0 * B * heads * N * N * hd / ms / 1e9:.1f} TFLOP/s useful")
    d = dp
    mask = bits
    dout = torch.ones_like(out)
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty_like(lse)
    dbias = torch.zeros_like(bias)
    for _ in range(2):
        lib.vsr_window_attention_bwd(ctypes.byref(d), VF._ptr(qkv), VF._ptr(bias), VF._ptr(mask), VF._ptr(dout), VF._ptr(lse), VF._ptr(delta),
                                     VF._ptr(dqkv), VF._ptr(dbias), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.vsr_window_attention_bwd(ctypes.byref(d), VF._ptr(qkv), VF._ptr(bias), VF._ptr(mask), VF._ptr(dout), VF._ptr(lse), VF._ptr(delta),
                                     VF._ptr(dqkv), VF._ptr(dbias), st)
    e1.record()
    torch.cuda.synchronize()
    print(f"bwd bias+mask: {e0.elapsed_time(e1) / iters:.3f} ms")


if __name__ == "__main__":
    main()

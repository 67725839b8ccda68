"""Micro-benchmark of the dominant kernels at the 540p trunk shape (for rocprofv3 / PMC runs).
    python tools/bench_kernels.py [conv|wgrad|all] [iters]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    h, w = 540, 960
    import vsrlab_amd
    from vsrlab_amd import functional as VF
    lib = vsrlab_amd._lib.load()
    dev = torch.device("cuda:0")
    x = torch.randn(1, h, (w + 31) // 32, 8, 32, 8, device=dev).to(torch.bfloat16)
    r = torch.randn(1, h, (w + 31) // 32, 8, 32, 8, device=dev).to(torch.bfloat16)
    if os.environ.get("BENCH_ZEROS") == "1":      # clock experiment: all-zero operands draw less power (MI355X_MICROARCH, DVFS give-back)
        x.zero_(); r.zero_()
    y = torch.empty_like(x)
    wgt = torch.randn(64, 64, 3, 3, device=dev) * (0.0 if os.environ.get("BENCH_ZEROS") == "1" else 0.04)
    b = torch.zeros(64, device=dev)
    wpack = torch.empty(9 * 64 * 64, dtype=torch.bfloat16, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    P = h * w
    if what in ("conv", "all"):
        for variant, res, act in (("bias+relu", None, 1), ("bias+res", r, 0)):
            for _ in range(3):
                lib.vsr_conv3x3_c64_fwd(1, VF._ptr(x), VF._ptr(wgt), VF._ptr(b), VF._ptr(wpack), VF._ptr(y), VF._ptr(res), act, 1, h, w, st)
            e0.record()
            for _ in range(iters):
                lib.vsr_conv3x3_c64_fwd(1, VF._ptr(x), VF._ptr(wgt), VF._ptr(b), VF._ptr(wpack), VF._ptr(y), VF._ptr(res), act, 1, h, w, st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / iters * 1e3
            print(f"conv3x3_c64 {variant}: {us:.1f} us  {2 * P * 64 * 576 / us / 1e6:.0f} TFLOP/s  {(128 + (64 if res is not None else 0)) * P * 2 / us / 1e3:.0f} GB/s")
    if what in ("wgrad", "all"):
        slab = torch.empty(lib.vsr_conv3x3_c64_wgrad_slab_floats(), dtype=torch.float32, device=dev)
        gw = torch.empty(64, 64, 3, 3, device=dev)
        gb = torch.empty(64, device=dev)
        for _ in range(2):
            lib.vsr_conv3x3_c64_wgrad(1, VF._ptr(x), VF._ptr(r), VF._ptr(gw), VF._ptr(gb), VF._ptr(slab), 1, h, w, st)
        e0.record()
        for _ in range(iters):
            lib.vsr_conv3x3_c64_wgrad(1, VF._ptr(x), VF._ptr(r), VF._ptr(gw), VF._ptr(gb), VF._ptr(slab), 1, h, w, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"wgrad3x3_c64 (1 frame, incl. reduce): {us:.1f} us  {2 * P * 64 * 576 / us / 1e6:.0f} TFLOP/s")


if __name__ == "__main__":
    main()

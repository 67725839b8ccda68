"""How much of a 540p conv3x3_c64 launch is start-up / drain: time per frame when N frames share one launch.
    python tools/bench_conv_batch.py [iters]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    h, w = 540, 960
    import vsrlab_amd
    from vsrlab_amd import functional as VF
    lib = vsrlab_amd._lib.load()
    dev = torch.device("cuda:0")
    wgt = torch.randn(64, 64, 3, 3, device=dev) * 0.04
    b = torch.zeros(64, device=dev)
    wpack = torch.empty(9 * 64 * 64, dtype=torch.bfloat16, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for n in (7, 4, 2, 1, 2, 4, 7):
        # 6 rotating buffer sets of n frames so that nothing is Infinity-Cache resident between launches
        sets = [(torch.randn(n, h, (w + 31) // 32, 8, 32, 8, device=dev).to(torch.bfloat16), torch.empty(n, h, (w + 31) // 32, 8, 32, 8, dtype=torch.bfloat16, device=dev))
                for _ in range(max(2, 12 // n))]
        for x, y in sets:
            lib.vsr_conv3x3_c64_fwd(1, VF._ptr(x), VF._ptr(wgt), VF._ptr(b), VF._ptr(wpack), VF._ptr(y), None, 1, n, h, w, st)
        e0.record()
        for i in range(iters):
            x, y = sets[i % len(sets)]
            lib.vsr_conv3x3_c64_fwd(1, VF._ptr(x), None, VF._ptr(b), VF._ptr(wpack), VF._ptr(y), None, 1, n, h, w, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"N = {n}: {us:.1f} us per launch, {us / n:.1f} us per frame", flush=True)


if __name__ == "__main__":
    main()

"""Interleaved A/B of builds of the persistent 3x3 64->64 conv kernel (cdna_hip_programming.md rule 24: variants x rounds in ONE
process, on random data, out of cache).

    python tools/ab_conv.py [rounds] lib_a.so lib_b.so ...      (paths relative to vsrlab_amd/lib/)

Every library is loaded side by side (ctypes); each round times `iters` launches of every library back to back: the two trunk
epilogues (bias+ReLU with sign-bit output is not reachable through this entry point, so: bias+ReLU / bias+skip) over 8 rotating
buffer sets (1.6 GB, beyond the Infinity Cache).  Diagnostic builds (make ABL=<bits>) also report the in-kernel clock
(s_memtime / s_memrealtime of the last launch, median over workgroups)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = ctypes.c_void_p


def main():
    rounds = int(sys.argv[1])
    names = sys.argv[2:]
    libs = [ctypes.CDLL(os.path.join(ROOT, "vsrlab_amd", "lib", n)) for n in names]
    h, w, nsets, iters = 540, 960, 8, 96
    dev = torch.device("cuda:0")
    shape = (1, h, (w + 31) // 32, 8, 32, 8)
    xs = [torch.randn(shape, device=dev).to(torch.bfloat16) for _ in range(nsets)]
    rs = [torch.randn(shape, device=dev).to(torch.bfloat16) for _ in range(nsets)]
    ys = [torch.empty_like(xs[0]) for _ in range(nsets)]
    wgt = torch.randn(64, 64, 3, 3, device=dev) * 0.04
    b = torch.zeros(64, device=dev)
    wpack = torch.empty(9 * 64 * 64, dtype=torch.bfloat16, device=dev)
    st = P(torch.cuda.current_stream().cuda_stream)
    libs[0].vsr_conv3x3_c64_fwd(1, P(xs[0].data_ptr()), P(wgt.data_ptr()), P(b.data_ptr()), P(wpack.data_ptr()), P(ys[0].data_ptr()), P(0), 1, 1, h, w, st)

    def launch(lib, i, mode):
        k = i % nsets
        res = mode == 1 or (mode == 2 and (i & 1))
        lib.vsr_conv3x3_c64_fwd(1, P(xs[k].data_ptr()), P(0), P(b.data_ptr()), P(wpack.data_ptr()), P(ys[k].data_ptr()),
                                P(rs[k].data_ptr() if res else 0), 0 if res else 1, 1, h, w, st)

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # ~2 s of back-to-back launches first: the clock settles under load (MI355X_MICROARCH, DVFS give-back item 6)
    for i in range(2000000 // 40):
        launch(libs[i % len(libs)], i, 2)
    torch.cuda.synchronize()
    res = {n: {0: [], 1: []} for n in names}
    clk = {n: {0: [], 1: []} for n in names}
    for r in range(rounds):
        for mode in (0, 1):
            for n, lib in zip(names, libs):
                for i in range(nsets):
                    launch(lib, i, mode)
                e0.record()
                for i in range(iters):
                    launch(lib, i, mode)
                e1.record()
                torch.cuda.synchronize()
                res[n][mode].append(e0.elapsed_time(e1) / iters * 1e3)
                if hasattr(lib, "vsr_debug_read_clk"):
                    out = np.zeros(256 * 4, dtype=np.uint64)
                    if lib.vsr_debug_read_clk(out.ctypes.data_as(P)) == 0:
                        o = out.reshape(256, 4).astype(np.float64)
                        ok = o[:, 1] > 0
                        clk[n][mode].append((float(np.median(o[ok, 0] / o[ok, 1] * 0.1)), float(np.median(o[ok, 0])), float(o[ok, 0].max())))
    for n in names:
        a0, a1 = np.array(res[n][0]), np.array(res[n][1])
        line = f"{n:32s} bias+relu med {np.median(a0):6.2f} min {a0.min():6.2f} | bias+skip med {np.median(a1):6.2f} min {a1.min():6.2f} us"
        for mode, nm in ((0, "relu"), (1, "skip")):
            if clk[n][mode]:
                c = np.array(clk[n][mode])
                line += f" | {nm}: clock {np.median(c[:, 0]):.3f} GHz, {np.median(c[:, 1]) / 1e3:.1f} k cycles per workgroup (median; slowest {np.median(c[:, 2]) / 1e3:.1f} k)"
        print(line, flush=True)


if __name__ == "__main__":
    main()

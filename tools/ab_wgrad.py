"""Interleaved A/B of builds of the 3x3 64->64 weight-gradient kernel (7 frames of 540x960 per launch, as inside a step).

    python tools/ab_wgrad.py [rounds] lib_a.so lib_b.so ...      (paths relative to vsrlab_amd/lib/)"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = ctypes.c_void_p


def main():
    rounds = int(sys.argv[1])
    names = sys.argv[2:]
    libs = [ctypes.CDLL(os.path.join(ROOT, "vsrlab_amd", "lib", n)) for n in names]
    for lib in libs:
        lib.vsr_conv3x3_c64_wgrad_slab_floats.restype = ctypes.c_size_t
    h, w, nf, nsets, iters = 540, 960, 7, 3, 12
    dev = torch.device("cuda:0")
    shape = (nf, h, (w + 31) // 32, 8, 32, 8)
    xs = [torch.randn(shape, device=dev).to(torch.bfloat16) for _ in range(nsets)]
    ds = [torch.randn(shape, device=dev).to(torch.bfloat16) for _ in range(nsets)]
    slab = torch.empty(libs[0].vsr_conv3x3_c64_wgrad_slab_floats(), dtype=torch.float32, device=dev)
    gw = torch.empty(64, 64, 3, 3, device=dev)
    gb = torch.empty(64, device=dev)
    st = P(torch.cuda.current_stream().cuda_stream)

    def launch(lib, i):
        k = i % nsets
        rc = lib.vsr_conv3x3_c64_wgrad(1, P(xs[k].data_ptr()), P(ds[k].data_ptr()), P(gw.data_ptr()), P(gb.data_ptr()), P(slab.data_ptr()), nf, h, w, st)
        assert rc == 0, rc

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(60):
        launch(libs[i % len(libs)], i)
    torch.cuda.synchronize()
    res = {n: [] for n in names}
    clk = {}
    extra = {}
    for r in range(rounds):
        for n, lib in zip(names, libs):
            for i in range(nsets):
                launch(lib, i)
            e0.record()
            for i in range(iters):
                launch(lib, i)
            e1.record()
            torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / iters * 1e3)
            if hasattr(lib, "vsr_debug_read_wclk"):
                out = np.zeros(1024, dtype=np.uint64)
                if lib.vsr_debug_read_wclk(out.ctypes.data_as(P)) == 0:
                    o = out.reshape(256, 4).astype(np.float64)
                    ok = o[:, 1] > 0
                    clk.setdefault(n, []).append(float(np.median(o[ok, 0] / o[ok, 1] * 0.1)))
                    extra[n] = (float(np.median(o[ok, 0])), float(np.median(o[ok, 2])), float(np.median(o[ok, 3])))
    for n in names:
        a = np.array(res[n])
        print(f"{n:40s} 7-frame wgrad + reduce: med {np.median(a):7.1f} min {a.min():7.1f} us  ({np.median(a) / nf:.1f} us per frame-conv)"
              + (f"  in-kernel clock {np.median(clk[n]):.3f} GHz; per workgroup {extra[n][0] / 1e3:.1f} k cycles, {extra[n][2]:.0f} tiles, consumer barrier wait {extra[n][1] / 1e3:.1f} k cycles" if n in clk else ""), flush=True)


if __name__ == "__main__":
    main()

"""Interleaved A/B of builds of the persistent 7x7 kernels (conv7x7_persist.hip) on the five SPyNet layer shapes at the finest
pyramid level of BASELINE config 2 (12 pairs x 544 x 960), through vsr_conv_layer_fwd (its weight pack, a few microseconds, is
inside the timed region).

    python tools/ab_c7.py [rounds] lib_a.so lib_b.so ...      (paths relative to vsrlab_amd/lib/)

Diagnostic builds (make ABL=<bits> ABLSRC=conv7x7_persist) also report the in-kernel clock of the last launch."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = ctypes.c_void_p
LAYERS = [(16, 8, 32, 32), (32, 32, 64, 64), (64, 64, 32, 32), (32, 32, 16, 16), (16, 16, 2, 0)]     # cin_pm, cin_real, cout_real, cd


def main():
    rounds = int(sys.argv[1])
    names = sys.argv[2:]
    libs = [ctypes.CDLL(os.path.join(ROOT, "vsrlab_amd", "lib", n)) for n in names]
    for lib in libs:
        lib.vsr_conv_layer_fwd.argtypes = [ctypes.c_int, ctypes.c_int, P, ctypes.c_int, P, P, P, ctypes.c_int, ctypes.c_int, P, P, ctypes.c_int, P,
                                           ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    n, h, w, iters = 12, 544, 960, 6
    dev = torch.device("cuda:0")
    st = P(torch.cuda.current_stream().cuda_stream)
    bufs = {c: [torch.randn((n, h, w // 32, c // 8, 32, 8), device=dev).to(torch.bfloat16) for _ in range(2)] for c in (16, 32, 64)}
    planar = torch.empty(n, 2, h, w, device=dev)
    wts = [torch.randn(co, ci, 7, 7, device=dev) * 0.02 for (_, ci, co, _) in LAYERS]
    bs = [torch.zeros(max(co, 4), device=dev) for (_, _, co, _) in LAYERS]
    wpack = torch.empty(49 * 64 * 64 * 4, dtype=torch.bfloat16, device=dev)

    def launch(lib, j, i):
        cp, ci, co, cd = LAYERS[j]
        x = bufs[cp][i & 1]
        y = bufs[cd][(i & 1) ^ 1] if cd else None
        rc = lib.vsr_conv_layer_fwd(1, 7, P(x.data_ptr()), cp, P(0), P(wts[j].data_ptr()), P(bs[j].data_ptr()), ci, co, P(wpack.data_ptr()),
                                    P(y.data_ptr() if cd else 0), cd, P(0 if cd else planar.data_ptr()), 1, 0.0, 0, n, h, w, st)
        assert rc == 0, rc

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(60):
        launch(libs[i % len(libs)], i % 5, i)
    torch.cuda.synchronize()
    res = {(nm, j): [] for nm in names for j in range(5)}
    clk = {(nm, j): [] for nm in names for j in range(5)}
    for r in range(rounds):
        for j in range(5):
            for nm, lib in zip(names, libs):
                launch(lib, j, 0)
                e0.record()
                for i in range(iters):
                    launch(lib, j, i)
                e1.record()
                torch.cuda.synchronize()
                res[(nm, j)].append(e0.elapsed_time(e1) / iters * 1e3)
                if hasattr(lib, "vsr_debug_read_clk7"):
                    out = np.zeros(256 * 4, dtype=np.uint64)
                    if lib.vsr_debug_read_clk7(out.ctypes.data_as(P)) == 0:
                        o = out.reshape(256, 4).astype(np.float64)
                        ok = o[:, 1] > 0
                        clk[(nm, j)].append((float(np.median(o[ok, 0] / o[ok, 1] * 0.1)), float(np.median(o[ok, 0]))))
    for j in range(5):
        for nm in names:
            a = np.array(res[(nm, j)])
            line = f"{LAYERS[j][1]:2d}->{LAYERS[j][2]:2d}  {nm:44s} med {np.median(a):8.1f} min {a.min():8.1f} us"
            if clk[(nm, j)]:
                c = np.array(clk[(nm, j)])
                line += f" | clock {np.median(c[:, 0]):.3f} GHz, {np.median(c[:, 1]) / 1e3:.1f} k cycles per workgroup"
            print(line, flush=True)


if __name__ == "__main__":
    main()

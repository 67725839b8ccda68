#!/usr/bin/env python3
"""The trunk-chain kernel (csrc/conv3x3_chain.hip) on bench.py's roofline-leg workload, for one or more builds of the library:

    python tools/ab_chain.py [lib.so ...]        (default: the ABL=0 diagnostic build, which carries the in-kernel clock)

Each build runs in a child process (the library path is fixed at import: VSRLAB_AMD_LIB).  Prints the launch time per layer
and, for diagnostic builds (make ABL=<bits> ABLSRC=conv3x3_chain), the in-kernel clock: s_memtime / s_memrealtime around the
whole kernel, median over workgroups -- cycles per workgroup and GHz (MI355X_MICROARCH "DVFS give-back" item 6)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import ctypes
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from vsrlab_amd import _lib, functional as VF
    import bench
    dev = torch.device("cuda:0")
    h, w, L, iters = 540, 960, bench.CHAIN_LAYERS, bench.CHAIN_ITERS
    g = torch.Generator(device="cpu").manual_seed(5)
    ws = [(torch.randn(64, 64, 3, 3, generator=g) * 0.04).to(dev) for _ in range(L)]
    ch = VF.ResidualChainC64(ws, [None] * L, 1, h, w, dev)
    ch.image(0).copy_(VF.to_pixel_major(torch.randn(1, 64, h, w, device=dev), VF.DT_BF16).reshape(-1))
    for _ in range(2):
        ch.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ch.launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    msg = f"{os.path.basename(_lib.LIB_PATH)} chain of {L} layers: {us:8.1f} us per launch = {us / L:6.2f} us per layer"
    lib = _lib.load()
    if hasattr(lib, "vsr_debug_read_clk_chain"):
        out = np.zeros((256, 2), dtype=np.uint64)
        if lib.vsr_debug_read_clk_chain(out.ctypes.data_as(ctypes.c_void_p)) == 0:
            cyc, tick = np.median(out[:, 0].astype(np.float64)), np.median(out[:, 1].astype(np.float64))
            msg += f" | chain: clock {cyc / tick / 10.0:.3f} GHz, {cyc / L / 1e3:.2f} k cycles per workgroup and layer (median)"
    print(msg, flush=True)


if __name__ == "__main__":
    if os.environ.get("AB_CHAIN_CHILD"):
        child()
        sys.exit(0)
    libs = sys.argv[1:] or [os.path.join(ROOT, "vsrlab_amd", "lib", "libvsrlab_hip_conv3x3_chain_abl0.so")]
    for lib in libs:
        env = dict(os.environ, AB_CHAIN_CHILD="1", VSRLAB_AMD_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env)
        if r.returncode != 0:
            sys.exit(r.returncode)

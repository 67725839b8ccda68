"""Diagnostic: per-phase cycle shares of the persistent conv kernel (needs `make -C vsrlab_amd/csrc STAMPS=1`)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "vsrlab_amd", "lib", "libvsrlab_hip_stamps.so"))
P = ctypes.c_void_p
h, w = 540, 960
dev = torch.device("cuda:0")
x = torch.randn(1, h, (w + 31) // 32, 8, 32, 8, device=dev).to(torch.bfloat16)
r = torch.randn(1, h, (w + 31) // 32, 8, 32, 8, device=dev).to(torch.bfloat16)
y = torch.empty_like(x)
wgt = torch.randn(64, 64, 3, 3, device=dev) * 0.04
b = torch.zeros(64, device=dev)
wpack = torch.empty(9 * 64 * 64, dtype=torch.bfloat16, device=dev)
st = P(torch.cuda.current_stream().cuda_stream)
for res, act, name in ((None, 1, "bias+relu"), (r, 0, "bias+res")):
    for _ in range(3):
        lib.vsr_conv3x3_c64_fwd(1, P(x.data_ptr()), P(wgt.data_ptr()), P(b.data_ptr()), P(wpack.data_ptr()), P(y.data_ptr()),
                                P(res.data_ptr() if res is not None else 0), act, 1, h, w, st)
    torch.cuda.synchronize()
    out = np.zeros(256 * 8 * 8, dtype=np.uint64)
    assert lib.vsr_debug_read_stamps(out.ctypes.data_as(P)) == 0
    s = out.reshape(256, 8, 8).astype(np.float64)
    print(f"[{name}] s_memtime ticks (100 MHz), mean over 256 CUs; whole loop incl. prologue wait: {s[:, :, 0].mean():.0f}")
    for role, waves, slots in (("MFMA waves", slice(0, 4), ((1, "epilogue operand loads"), (2, "K loop"), (3, "epilogue"), (4, "barrier"), (5, "prologue (once)"), (6, "  weight staging + set-up"))),
                               ("DMA waves", slice(4, 8), ((5, "issue next tile"), (6, "vmcnt(0)"), (7, "barrier"), (2, "prologue: start -> first tile issued"), (1, "  of which: until the issue starts"), (4, "  vmcnt(0)")))):
        for k, nm in slots:
            v = s[:, waves, k]
            print(f"   {role:10s} {nm:24s} mean {v.mean():8.0f}  min {v.min():8.0f}  max {v.max():8.0f}")

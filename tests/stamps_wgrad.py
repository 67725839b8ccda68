"""Diagnostic: per-phase cycle shares of the DMA weight-gradient kernel (needs `make -C vsrlab_amd/csrc STAMPS=1`)."""
import ctypes
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "vsrlab_amd", "lib", "libvsrlab_hip_stamps.so"))
lib.vsr_conv3x3_c64_wgrad_slab_floats.restype = ctypes.c_size_t
P = ctypes.c_void_p
h, w = 540, 960
dev = torch.device("cuda:0")
x = torch.randn(1, h, (w + 31) // 32, 8, 32, 8, device=dev).to(torch.bfloat16)
r = torch.randn(1, h, (w + 31) // 32, 8, 32, 8, device=dev).to(torch.bfloat16)
slab = torch.empty(lib.vsr_conv3x3_c64_wgrad_slab_floats(), dtype=torch.float32, device=dev)
gw = torch.empty(64, 64, 3, 3, device=dev)
gb = torch.empty(64, device=dev)
st = P(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    lib.vsr_conv3x3_c64_wgrad(1, P(x.data_ptr()), P(r.data_ptr()), P(gw.data_ptr()), P(gb.data_ptr()), P(slab.data_ptr()), 1, h, w, st)
torch.cuda.synchronize()
out = np.zeros(256 * 8 * 8, dtype=np.uint64)
assert lib.vsr_debug_read_wgrad_stamps(out.ctypes.data_as(P)) == 0
s = out.reshape(256, 8, 8).astype(np.float64)
print(f"[wgrad, 1 frame = 8 tiles per CU] shader cycles, mean over 256 CUs x 8 waves; loop incl. first-tile wait: {s[:, :, 0].mean():.0f}")
for k, nm in ((1, "DMA issue (next tile)"), (2, "bias partial sums"), (3, "K loop"), (4, "vmcnt(0)"), (5, "barrier")):
    v = s[:, :, k]
    print(f"   {nm:24s} mean {v.mean():8.0f}  min {v.min():8.0f}  max {v.max():8.0f}")

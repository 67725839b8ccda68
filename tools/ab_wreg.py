#!/usr/bin/env python3
"""conv3x3_wreg.hip (weights in registers, VSRLAB_AMD_WREG=1) against conv3x3_persist.hip: bit-identity on a ragged and a 540p image, then
the one-launch-per-layer roofline leg of bench.py for both, interleaved (vsr_debug_set_wreg flips it between launches)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vsrlab_amd import functional as VF, _lib
import bench
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
for (n, h, w) in [(2, 21, 70), (1, 96, 160), (1, 540, 960)]:
    x = VF.to_pixel_major(torch.randn(n, 64, h, w, generator=g).to(dev), VF.DT_BF16)
    r = VF.to_pixel_major(torch.randn(n, 64, h, w, generator=g).to(dev), VF.DT_BF16)
    wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.04).to(dev)
    b = (torch.randn(64, generator=g) * 0.1).to(dev)
    out = {}
    for m in ("0", "1"):
        _lib.load().vsr_debug_set_wreg(int(m))
        out[m] = (VF.from_pixel_major(VF.conv3x3_c64(x, wt, b, act=1)), VF.from_pixel_major(VF.conv3x3_c64(x, wt, b, act=0, res_pm=r)))
    torch.cuda.synchronize()
    print((n, h, w), "relu identical:", torch.equal(out["0"][0], out["1"][0]), " skip identical:", torch.equal(out["0"][1], out["1"][1]),
          float((out["0"][0] - out["1"][0]).abs().max()), float((out["0"][1] - out["1"][1]).abs().max()), flush=True)
for rep in range(3):
    for m in ("0", "1"):
        _lib.load().vsr_debug_set_wreg(int(m))
        leg = bench.per_layer_kernel_roofline(dev, 540, 960)
        print("WREG", m, leg["avg_us"], "us per launch,", leg["frac"], flush=True)

"""Prints (does not assert) the error metrics of the HIP BasicVSR path against the oracle, used
to set and justify the tolerances in test_hip_parity.py.  Run on the GPU box:
    python tools/gpu_diag.py [rb] [t] [h] [w]
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
from helpers import rand, rel_err, rel_l2  # noqa: E402
from oracle import basicvsr_oracle as O  # noqa: E402


def main():
    rb = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    t = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    w = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    n = 2
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.basicvsr import BasicVSR
    dev = torch.device("cuda:0")
    sd = O.keyed_state_dict(O.basicvsr_param_shapes(64, rb, 4))
    lrs = rand(10, n, t, 3, h, w)
    cot = rand(13, n, t, 3, 4 * h, 4 * w, lo=-1, hi=1)
    hr = rand(11, n, t, 3, 4 * h, 4 * w)
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    t0 = time.time()
    sr32, _, g32 = O.fwd_bwd(sd, lrs, hr, cot=cot)
    print(f"oracle fp32: {time.time() - t0:.1f}s on {torch.get_num_threads()} threads")
    t0 = time.time()
    with O.emulate_bf16():
        srq, _, gq = O.fwd_bwd(sd, lrs, hr, cot=cot)
    print(f"oracle bf16-emulation: {time.time() - t0:.1f}s; vs fp32 oracle: sr {rel_err(srq, sr32):.2e}")
    for dtype, sr_o, g_o in (("fp32", sr32, g32), ("bf16", srq, gq), ("bf16-vs-fp32oracle", sr32, g32)):
        m = BasicVSR(64, rb, 4, False, False)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev)
        m.compute_dtype = dtype.split("-")[0]
        sr = m(lrs.to(dev))
        torch.mean(sr * cot.to(dev)).backward()
        grads = {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}
        per = sorted(((rel_l2(grads[k], g_o[k]), k) for k in g_o), reverse=True)
        allg = torch.cat([grads[k].flatten() for k in g_o])
        allo = torch.cat([g_o[k].flatten() for k in g_o])
        print(f"[{dtype}] sr max-rel {rel_err(sr, sr_o):.3e} rel-l2 {rel_l2(sr, sr_o):.3e} | grads global rel-l2 {rel_l2(allg, allo):.3e} "
              f"| worst per-tensor {per[0][0]:.3e} {per[0][1]} | median {per[len(per) // 2][0]:.3e}")
        print("    top5:", [(f"{e:.2e}", k) for e, k in per[:5]])
        if os.environ.get("DIAG_KEYS"):
            order = ["conv_last.2.weight", "conv_last.2.bias", "conv_last.0.weight", "conv_last.0.bias", "upsample.1.upconv.weight",
                     "upsample.1.upconv.bias", "upsample.0.upconv.weight", "upsample.0.upconv.bias", "point_conv.0.weight",
                     "point_conv.0.bias"]
            for d in ("forward_resblocks", "backward_resblocks"):
                for b in range(rb - 1, -1, -1):
                    order += [f"{d}.res_block.{b}.conv2.weight", f"{d}.res_block.{b}.conv2.bias", f"{d}.res_block.{b}.conv1.weight",
                              f"{d}.res_block.{b}.conv1.bias"]
                order += [f"{d}.conv.0.weight", f"{d}.conv.0.bias"]
            for k in order:
                print(f"      {k:48s} rel-l2 {rel_l2(grads[k], g_o[k]):.3e}  max-rel {rel_err(grads[k], g_o[k]):.3e}")


if __name__ == "__main__":
    main()

"""BASELINE config 3 at size, for timing and rocprofv3: one GAN iteration = generator_step + backward + FusedAdam,
discriminator_step + backward + FusedAdam (train_gan.py:35-58) on a (1,7,3,540,960) clip / seven 2160x3840 frames, bf16.
    python tools/bench_gan.py [iters] [d|g|all]      d = discriminator step only, g = generator step only"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VSRLAB_AMD_DTYPE", "bf16")


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    what = sys.argv[2] if len(sys.argv) > 2 else "all"
    import importlib
    from vsrlab_amd.core.losses import AdversarialLoss, CharbonnierLoss
    from vsrlab_amd.optim import FusedAdam
    from vsrlab_amd.train_gan import discriminator_step, dummy_loss, generator_step
    from vsrlab_amd.vsr.models.RealBasicVSR.realbasicvsr import RealBasicVSR
    UNetDiscriminator = importlib.import_module("vsrlab_amd.vsr.models.RealBasicVSR.modules.unet-discriminator").UNetDiscriminator
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    g = RealBasicVSR(20, mid_channels=64, upscale=4, res_blocks=20, pretrained_flow=False, train_flow=False).to(dev)
    g.basicvsr.compute_dtype = "bf16"
    d = UNetDiscriminator(3, 64).to(dev).train()
    d.compute_dtype = "bf16"
    opt_g = FusedAdam(g.parameters(), lr=1e-4, betas=(0.9, 0.99))
    opt_d = FusedAdam(d.parameters(), lr=1e-4, betas=(0.9, 0.99))
    lr = torch.rand(1, 7, 3, 540, 960, device=dev)
    hr = torch.rand(1, 7, 3, 2160, 3840, device=dev)
    adv, crit = AdversarialLoss(), CharbonnierLoss()
    sr_fixed = torch.rand(1, 7, 3, 2160, 3840, device=dev)
    for it in range(iters + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sr = sr_fixed
        if what in ("all", "g"):
            sr, loss_g, _, _ = generator_step(g, d, crit, dummy_loss, adv, lr, hr)
            loss_g.backward()
            opt_g.step(max_grad_norm=1.0)
            opt_g.zero_grad()
            del loss_g
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if what in ("all", "d"):
            loss_d = discriminator_step(d, adv, sr, hr)
            loss_d.backward()
            opt_d.step(max_grad_norm=1.0)
            opt_d.zero_grad()
            del loss_d
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"iter {it}: generator step {(t1 - t0) * 1e3:.0f} ms, discriminator step {(t2 - t1) * 1e3:.0f} ms, "
              f"total {(t2 - t0) * 1e3:.0f} ms, peak {torch.cuda.max_memory_allocated() / 2**30:.0f} GiB", flush=True)


if __name__ == "__main__":
    main()

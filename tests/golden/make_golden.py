"""Generate the golden vectors in this directory from the REAL reference.

Run ONLY in the build container (needs /root/reference, which never travels):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference ships no tests or fixtures of its own, so the
files written here are the only pin for the oracle and, through it, for the HIP
path.  Only inputs' seeds and the reference's OUTPUTS are stored (fp32 .npz);
weights are regenerated on both sides from ``oracle.basicvsr_oracle.keyed_tensor``
(a pure function of the state_dict key), so nothing of the reference's source or
parameters is committed.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF_SRC = "/root/reference/src"


def import_reference():
    """`vsrlab` -> /root/reference/src (the reference's setup.py:6-7 alias) plus a stub
    for the unused ``torchvision.ops`` import at core/modules/conv.py:4."""
    pkg = types.ModuleType("vsrlab")
    pkg.__path__ = [REF_SRC]
    sys.modules["vsrlab"] = pkg
    tv = types.ModuleType("torchvision")
    ops = types.ModuleType("torchvision.ops")

    class DeformConv2d(nn.Module):
        pass

    ops.DeformConv2d = DeformConv2d
    ops.deform_conv2d = None
    tv.ops = ops
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = ops
    from vsrlab.vsr.models.RealBasicVSR.modules import basicvsr, spynet
    from vsrlab.vsr.models.RealBasicVSR import realbasicvsr
    from vsrlab.core.modules import conv, upsampling
    return basicvsr, spynet, realbasicvsr, conv, upsampling


def load_keyed(module):
    from oracle.basicvsr_oracle import keyed_tensor
    sd = {k: keyed_tensor(k, tuple(v.shape)) for k, v in module.state_dict().items()}
    module.load_state_dict(sd, strict=True)
    return module


def rand(seed, *shape, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def save(name, **arrs):
    out = {k: (v.detach().numpy().astype(np.float32) if isinstance(v, torch.Tensor) else np.asarray(v))
           for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    basicvsr, spynet, realbasicvsr, conv, upsampling = import_reference()

    # ---- (i) per-op -------------------------------------------------------------
    x = rand(1, 2, 5, 9, 11, lo=-1, hi=1)
    flow = rand(2, 2, 2, 9, 11, lo=-8, hi=8)       # +-8 px incl. out-of-range taps
    fl = flow.permute(0, 2, 3, 1)
    save("flow_warp", seed_x=1, seed_flow=2,
         zeros=spynet.flow_warp(x, fl), border=spynet.flow_warp(x, fl, padding_mode="border"))

    rc = load_keyed(conv.ResidualConv(16))
    xi = rand(3, 2, 16, 12, 20, lo=-1, hi=1)
    save("residual_conv", seed_x=3, y=rc(xi))

    rb = load_keyed(conv.ResidualBlock(19, 16, 2))
    xi = rand(4, 2, 19, 12, 20, lo=-1, hi=1)
    save("residual_block", seed_x=4, y=rb(xi))

    ps = load_keyed(upsampling.PixelShufflePack(16, 16, 2))
    xi = rand(5, 2, 16, 7, 9, lo=-1, hi=1)
    save("pixel_shuffle_pack", seed_x=5, y=ps(xi))

    # ---- (ii) SPyNet: 40x72 exercises the resize-to-/32 path; 32x32 the 1x1 level --
    sp = load_keyed(spynet.Spynet(False))
    a = rand(6, 2, 3, 40, 72)
    b = rand(7, 2, 3, 40, 72)
    a2 = rand(8, 1, 3, 32, 32)
    b2 = rand(9, 1, 3, 32, 32)
    with torch.no_grad():
        save("spynet", seed_ref=6, seed_supp=7, flow=sp(a, b), seed_ref2=8, seed_supp2=9, flow2=sp(a2, b2))

    # ---- (iii) end to end: sr, Charbonnier loss, selected grads --------------------
    from oracle.basicvsr_oracle import charbonnier as _unused  # noqa: F401  (formula restated in-line below)
    for tag, mid, blocks, shape in (("basicvsr_m16_rb2", 16, 2, (1, 3, 3, 32, 32)),
                                    ("basicvsr_m64_rb3", 64, 3, (2, 3, 3, 24, 40))):
        # Run the reference in float64: LeakyReLU/ReLU masks are discontinuous, so two fp32
        # runs of the same maths differ by ~1e-3 on some weight grads (measured: reference fp32
        # vs reference fp64 = 1.7e-3 on conv_last.0.weight).  fp64 values are the stable pin.
        m = load_keyed(basicvsr.BasicVSR(mid, blocks, 4, False, False)).double()
        lrs = rand(10, *shape).double()
        n, t, _, h, w = shape
        hr = rand(11, n, t, 3, 4 * h, 4 * w).double()
        sr = m(lrs)
        d = sr - hr
        loss = torch.mean(torch.sqrt(d * d + 1e-9))     # core/losses.py:15-17
        # Gradients are pinned through a LINEAR functional mean(sr*cot): Charbonnier's own
        # gradient d/sqrt(d^2+1e-9) flips sign on |d|~1e-5 and is ill-conditioned for parity.
        cot = rand(13, n, t, 3, 4 * h, 4 * w, lo=-1, hi=1).double()
        torch.mean(sr * cot).backward()
        named = dict(m.named_parameters())
        keys = ["backward_resblocks.conv.0.weight", "backward_resblocks.conv.0.bias",
                f"forward_resblocks.res_block.{blocks - 1}.conv2.weight",
                "forward_resblocks.res_block.0.conv1.weight", "backward_resblocks.res_block.0.conv1.bias",
                "point_conv.0.weight", "upsample.0.upconv.weight", "upsample.1.upconv.bias",
                "conv_last.0.weight", "conv_last.2.weight", "conv_last.2.bias"]
        grads = {"grad__" + k.replace(".", "__"): named[k].grad for k in keys}
        with torch.no_grad():
            ff, fb = m.compute_flow(lrs)
        save(tag, seed_lr=10, seed_hr=11, seed_cot=13, sr=sr, loss=loss, flow_forward=ff, flow_backward=fb, **grads)

    # ---- RealBasicVSR forward (cleaner); reference mutates its input, so pass a clone ----
    m = load_keyed(realbasicvsr.RealBasicVSR(2, mid_channels=16, upscale=4, res_blocks=2,
                                             pretrained_flow=False, train_flow=False))
    lrs = rand(12, 1, 3, 3, 32, 32)
    with torch.no_grad():
        sr, lq = m(lrs.clone())
    save("realbasicvsr_m16", seed_lr=12, sr=sr, lq=lq)

    # the same at 64 channels (the width the HIP path is built for), ragged frame size
    m = load_keyed(realbasicvsr.RealBasicVSR(2, mid_channels=64, upscale=4, res_blocks=2,
                                             pretrained_flow=False, train_flow=False))
    lrs = rand(14, 1, 3, 3, 24, 40)
    with torch.no_grad():
        sr, lq = m(lrs.clone())
    save("realbasicvsr_m64", seed_lr=14, sr=sr, lq=lq)


def proj_vector(key, shape):
    """Seeded random direction for a parameter (a function of its state_dict key): gradients too large to store
    are pinned by <grad, proj_vector>, their sum and their L2 norm.  tests/helpers.py holds the same function."""
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(("proj:" + key).encode()) & 0x7fffffff)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


def trainflow():
    """BasicVSR with train_flow=True (conf/experiment/basic.yaml:7): the gradients of all 60 SPyNet tensors, through
    the propagation warps' flow gradient, the pyramid's border warps and x2 upsampling (spynet.py:38-106)."""
    torch.set_num_threads(8)
    basicvsr, spynet, realbasicvsr, conv, upsampling = import_reference()
    mid, blocks, shape = 64, 3, (2, 3, 3, 24, 40)               # same clip / weights / cotangent as basicvsr_m64_rb3
    m = load_keyed(basicvsr.BasicVSR(mid, blocks, 4, False, True)).double()
    lrs = rand(10, *shape).double()
    n, t, _, h, w = shape
    sr = m(lrs)
    cot = rand(13, n, t, 3, 4 * h, 4 * w, lo=-1, hi=1).double()
    torch.mean(sr * cot).backward()
    named = dict(m.named_parameters())
    spy = [k for k in named if k.startswith("spynet.")]
    assert len(spy) == 60 and all(named[k].grad is not None for k in spy)
    stats = torch.stack([torch.stack([named[k].grad.sum(), named[k].grad.norm(),
                                      (named[k].grad * proj_vector(k, named[k].shape)).sum()]) for k in spy])
    full = ["spynet.basic_module.5.basic_module.0.conv.0.weight", "spynet.basic_module.5.basic_module.4.conv.0.weight",
            "spynet.basic_module.5.basic_module.4.conv.0.bias", "spynet.basic_module.3.basic_module.1.conv.0.bias",
            "spynet.basic_module.2.basic_module.3.conv.0.weight", "spynet.basic_module.0.basic_module.4.conv.0.weight",
            "spynet.basic_module.0.basic_module.0.conv.0.bias"]
    grads = {"grad__" + k.replace(".", "__"): named[k].grad for k in full}
    trunk = ["backward_resblocks.conv.0.weight", "conv_last.2.weight"]     # unchanged by train_flow: cross-check
    grads.update({"grad__" + k.replace(".", "__"): named[k].grad for k in trunk})
    out = {k: v.detach().numpy().astype(np.float64) for k, v in grads.items()}
    out["spy_stats"] = stats.detach().numpy().astype(np.float64)
    out["spy_keys"] = np.array(spy)
    out["seed_lr"] = np.asarray(10); out["seed_cot"] = np.asarray(13)
    np.savez_compressed(os.path.join(HERE, "basicvsr_m64_rb3_trainflow.npz"), **out)
    print("basicvsr_m64_rb3_trainflow", {k: v.shape for k, v in out.items()})


def lrgrad():
    """Gradient of BasicVSR w.r.t. its input clip (what RealBasicVSR's pre-clean stack receives, realbasicvsr.py:11-15):
    through the stem's LR channels, the bilinear x4 skip (basicvsr.py:22,82), the propagation warps' flows and SPyNet's
    image pyramid (spynet.py:38-93, frozen weights)."""
    torch.set_num_threads(8)
    basicvsr, spynet, realbasicvsr, conv, upsampling = import_reference()
    mid, blocks, shape = 64, 3, (2, 3, 3, 24, 40)               # same clip / weights / cotangent as basicvsr_m64_rb3
    m = load_keyed(basicvsr.BasicVSR(mid, blocks, 4, False, False)).double()
    lrs = rand(10, *shape).double().requires_grad_(True)
    n, t, _, h, w = shape
    sr = m(lrs)
    cot = rand(13, n, t, 3, 4 * h, 4 * w, lo=-1, hi=1).double()
    torch.mean(sr * cot).backward()
    named = dict(m.named_parameters())
    out = {"grad_lrs": lrs.grad.detach().numpy().astype(np.float64),
           "grad__conv_last__2__weight": named["conv_last.2.weight"].grad.detach().numpy().astype(np.float64),
           "seed_lr": np.asarray(10), "seed_cot": np.asarray(13)}
    np.savez_compressed(os.path.join(HERE, "basicvsr_m64_rb3_lrgrad.npz"), **out)
    print("basicvsr_m64_rb3_lrgrad", {k: v.shape for k, v in out.items()})


def up2():
    """BasicVSR with upscale = 2 (the constructor is generic, basicvsr.py:12-23: upscale // 2 = ONE PixelShufflePack, conv_last at
    2h x 2w, `nn.Upsample(scale_factor=2)` skip; conf/train/model/basicvsr.yaml:4 takes the scale from the dataset config), float64:
    sr, every trainable gradient of mean(sr * cot), and the gradient w.r.t. the clip (the x2 bilinear adjoint among others)."""
    torch.set_num_threads(8)
    basicvsr, spynet, realbasicvsr, conv, upsampling = import_reference()
    mid, blocks, shape = 64, 3, (2, 3, 3, 24, 40)
    m = load_keyed(basicvsr.BasicVSR(mid, blocks, 2, False, False)).double()
    lrs = rand(10, *shape).double().requires_grad_(True)
    n, t, _, h, w = shape
    sr = m(lrs)
    cot = rand(13, n, t, 3, 2 * h, 2 * w, lo=-1, hi=1).double()
    torch.mean(sr * cot).backward()
    out = {"sr": sr.detach().numpy().astype(np.float32), "grad_lrs": lrs.grad.detach().numpy().astype(np.float64),
           "seed_lr": np.asarray(10), "seed_cot": np.asarray(13)}
    for k, prm in m.named_parameters():
        if prm.grad is not None:
            out["grad__" + k.replace(".", "__")] = prm.grad.detach().numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "basicvsr_m64_rb3_up2.npz"), **out)
    print("basicvsr_m64_rb3_up2", len(out), "arrays; sr", out["sr"].shape)


def realtrain():
    """RealBasicVSR training gradients.  The reference's own backward raises in fp32 (`x += residues` overwrites a tensor
    saved for backward, realbasicvsr.py:29); it runs under autocast, where the saved tensors are
    the bf16 copies -- which is how the reference is trained (train.py:93).  So this fixture is the reference under CPU
    autocast(bfloat16): a LOOSE pin (bf16 rounding) of the pre-clean stack's gradients; the sharp pin is the oracle."""
    torch.set_num_threads(8)
    basicvsr, spynet, realbasicvsr, conv, upsampling = import_reference()
    m = load_keyed(realbasicvsr.RealBasicVSR(2, mid_channels=64, upscale=4, res_blocks=2, pretrained_flow=False, train_flow=False))
    shape = (1, 3, 3, 24, 40)
    n, t, _, h, w = shape
    lr = rand(14, *shape)
    cot_sr = rand(15, n, t, 3, 4 * h, 4 * w, lo=-1, hi=1)
    cot_lq = rand(16, n, t, 3, h, w, lo=-1, hi=1)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        sr, lq = m(lr.clone())
    (torch.mean(sr.float() * cot_sr) + torch.mean(lq.float() * cot_lq)).backward()
    named = dict(m.named_parameters())
    keys = ["cleaner.resblock.conv.0.weight", "cleaner.resblock.res_block.0.conv1.weight", "cleaner.resblock.res_block.1.conv2.weight",
            "cleaner.conv.weight", "cleaner.conv.bias", "basicvsr.conv_last.2.weight", "basicvsr.backward_resblocks.conv.0.weight"]
    out = {"grad__" + k.replace(".", "__"): named[k].grad.detach().float().numpy() for k in keys}
    out.update(seed_lr=np.asarray(14), seed_cot_sr=np.asarray(15), seed_cot_lq=np.asarray(16),
               sr=sr.detach().float().numpy(), lq=lq.detach().float().numpy())
    np.savez_compressed(os.path.join(HERE, "realbasicvsr_m64_train_autocast.npz"), **out)
    print("realbasicvsr_m64_train_autocast", {k: v.shape for k, v in out.items()})


def disc():
    """UNetDiscriminator (unet-discriminator.py:4-31, SpectralConv core/modules/conv.py:6-13) and the losses of one GAN
    iteration with perceptual_loss = null (train_gan.py:35-58, core/losses.py:66-74, core/utils.py:235-240).  The module
    name has a hyphen, so it is imported the way Hydra does it (importlib.import_module, core/utils.py:138).  fp64
    reference, training mode (one power iteration per forward).  `kornia` is not installed: compute_loss's
    `resize` is kornia's documented default (bilinear, align_corners=False, no antialias) = F.interpolate, restated in
    the generating code below -- that one call is therefore pinned by torch, not by the reference ("parity unpinned"
    for the choice of kornia defaults)."""
    import importlib
    torch.set_num_threads(8)
    import_reference()
    mod = importlib.import_module("vsrlab.vsr.models.RealBasicVSR.modules.unet-discriminator")
    from oracle.basicvsr_oracle import keyed_tensor
    d = mod.UNetDiscriminator(3, 64)
    sd = {k: keyed_tensor(k, tuple(v.shape)) for k, v in d.state_dict().items()}
    d.load_state_dict(sd, strict=True)
    d = d.double().train()
    keys = sorted(d.state_dict().keys())
    img = rand(41, 2, 3, 32, 48).double().requires_grad_(True)
    cot = rand(42, 2, 1, 32, 48, lo=-1, hi=1).double()
    out = d(img)
    torch.mean(out * cot).backward()
    named = dict(d.named_parameters())
    store = {"seed_img": np.asarray(41), "seed_cot": np.asarray(42), "out": out.detach().float().numpy(),
             "dimg": img.grad.detach().float().numpy(), "keys": np.asarray(keys)}
    for k, p in named.items():
        g = p.grad.detach()
        tag = k.replace(".", "__")
        if g.numel() <= 40000:
            store["grad__" + tag] = g.float().numpy()
        store["gsum__" + tag] = np.asarray(float(g.sum()))
        store["gnorm__" + tag] = np.asarray(float(g.norm()))
        store["gproj__" + tag] = np.asarray(float((g * proj_vector(k, tuple(g.shape))).sum()))
    for k, v in d.state_dict().items():                      # buffers after the training-mode forward
        if k.endswith(("weight_u", "weight_v")):
            store["buf__" + k.replace(".", "__")] = v.detach().float().numpy()
    np.savez_compressed(os.path.join(HERE, "unet_discriminator.npz"), **store)
    print("unet_discriminator", len(store))

    # one GAN iteration's two losses and their gradients w.r.t. sr / lq (generator side) and D's parameters
    from vsrlab.core.modules import conv  # noqa: F401
    import torch.nn.functional as F

    class Adv(nn.Module):                                     # core/losses.py:66-74 (losses.py itself imports kornia / torchvision)
        def __init__(self, weight=2e-5):
            super().__init__()
            self.weight = weight

        def forward(self, x, target, is_disc=False):
            target = x.new_ones(x.size()) * target
            loss = F.binary_cross_entropy_with_logits(x, target)
            return loss if is_disc else loss * self.weight

    def charb(x, y, eps=1e-9):                                # core/losses.py:15-18
        diff = x - y
        return torch.mean(torch.sqrt((diff * diff) + eps))

    d.load_state_dict(sd, strict=True)
    d = d.double().train()
    adv = Adv()
    b, t, c, h, w = 1, 2, 3, 32, 48
    sr = rand(43, b, t, c, h, w).double().requires_grad_(True)
    hr = rand(44, b, t, c, h, w).double()
    lq = rand(45, b, t, c, h // 4, w // 4).double().requires_grad_(True)
    tgt = F.interpolate(hr.reshape(-1, c, h, w), size=(h // 4, w // 4), mode="bilinear", align_corners=False).reshape(b, t, c, h // 4, w // 4)
    pixel = charb(sr, hr) + charb(lq, tgt)                    # compute_loss, core/utils.py:235-240
    disc_sr = d(sr.reshape(-1, c, h, w))                      # generator_step, train_gan.py:35-48
    loss_g = pixel + 0.0 + adv(disc_sr, 1, False)
    loss_g.backward()
    g_sr, g_lq = sr.grad.detach().clone(), lq.grad.detach().clone()
    d.zero_grad()
    loss_d = adv(d(hr.reshape(-1, c, h, w)), 1, True) + adv(d(sr.detach().reshape(-1, c, h, w)), 0, True)   # train_gan.py:50-58
    loss_d.backward()
    store = {"seed_sr": np.asarray(43), "seed_hr": np.asarray(44), "seed_lq": np.asarray(45), "loss_g": np.asarray(float(loss_g)),
             "loss_d": np.asarray(float(loss_d)), "dsr": g_sr.float().numpy(), "dlq": g_lq.float().numpy()}
    for k, p in dict(d.named_parameters()).items():
        g = p.grad.detach()
        tag = k.replace(".", "__")
        store["gsum__" + tag] = np.asarray(float(g.sum()))
        store["gnorm__" + tag] = np.asarray(float(g.norm()))
        store["gproj__" + tag] = np.asarray(float((g * proj_vector(k, tuple(g.shape))).sum()))
    for k, v in d.state_dict().items():
        if k.endswith(("weight_u", "weight_v")):
            store["buf__" + k.replace(".", "__")] = v.detach().float().numpy()
    np.savez_compressed(os.path.join(HERE, "gan_step.npz"), **store)
    print("gan_step", len(store))


def vrt():
    """VRT window attention and one shifted TMSA block (vsr/models/VRT/modules/window_attention.py:100-188, :61-77;
    tmsa.py:9-124), fp64 reference.  dim 120 / 6 heads = head_dim 20, window (2,8,8) with mutual attention (the VRT
    stage configuration) and dim 180 / 6 = 30, window (6,8,8) without (the RTMSA configuration).  Parameters are keyed;
    the buffers (relative_position_index, position_bias) are the module's own deterministic values."""
    torch.set_num_threads(8)
    import_reference()
    from vsrlab.vsr.models.VRT.modules import tmsa as ref_tmsa
    from vsrlab.vsr.models.VRT.modules import window_attention as ref_wa
    from oracle.basicvsr_oracle import keyed_tensor

    def keyed_params(m):
        with torch.no_grad():
            for k, p in m.named_parameters():
                p.copy_(keyed_tensor(k, tuple(p.shape)))
        return m.double()

    store = {}

    def put_grads(tag, module):          # small gradients in full, every gradient by norm and seeded projection
        for k, p in module.named_parameters():
            g = p.grad.detach()
            name = k.replace(".", "__")
            if g.numel() <= 8192:
                store[f"{tag}__grad__{name}"] = g.float().numpy()
            store[f"{tag}__gnorm__{name}"] = np.asarray(float(g.norm()))
            store[f"{tag}__gproj__{name}"] = np.asarray(float((g * proj_vector(k, tuple(g.shape))).sum()))

    for tag, dim, ws, mut, B_ in (("a", 120, (2, 8, 8), True, 4), ("b", 180, (6, 8, 8), False, 2)):
        m = keyed_params(ref_wa.WindowAttention(dim, ws, 6, qkv_bias=True, qk_scale=None, mut_attn=mut))
        N = ws[0] * ws[1] * ws[2]
        x = rand(50 + ord(tag), B_, N, dim, lo=-1, hi=1).double().requires_grad_(True)
        mask = ref_wa.compute_mask(2 * ws[0], 16, 16, ws, tuple(i // 2 for i in ws), "cpu")[:2].double() if tag == "a" else None
        y = m(x, mask)
        cot = rand(60 + ord(tag), B_, N, dim, lo=-1, hi=1).double()
        (y * cot).sum().backward()
        store[f"{tag}__out"] = y.detach().float().numpy()
        store[f"{tag}__dx"] = x.grad.detach().float().numpy()
        put_grads(tag, m)
        store[f"{tag}__seed_x"], store[f"{tag}__seed_cot"] = np.asarray(50 + ord(tag)), np.asarray(60 + ord(tag))
    # one shifted TMSA block on a volume that needs padding in H (20 -> 24) and has 2 x 3 x 2 windows
    blk = keyed_params(ref_tmsa.TMSA(120, (4, 20, 16), 6, window_size=(2, 8, 8), shift_size=(1, 4, 4), mut_attn=True, mlp_ratio=2.,
                                     qkv_bias=True))
    x = rand(70, 1, 4, 20, 16, 120, lo=-1, hi=1).double().requires_grad_(True)
    mask = ref_wa.compute_mask(4, 24, 16, (2, 8, 8), (1, 4, 4), "cpu").double()
    y = blk(x, mask)
    cot = rand(71, 1, 4, 20, 16, 120, lo=-1, hi=1).double()
    (y * cot).sum().backward()
    store["t__out"], store["t__dx"] = y.detach().float().numpy(), x.grad.detach().float().numpy()
    store["t__mask"] = mask.float().numpy()
    put_grads("t", blk)
    store["t__keys"] = np.asarray(sorted(blk.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, "vrt_window_attention.npz"), **store)
    print("vrt_window_attention", len(store))
    # the VRT tree's canonical SpyNet (conf/train/model/spynet.yaml: return_levels [2,3,4,5]); pretrained weights are absent
    # (.MISSING_LARGE_BLOBS) -> keyed weights
    from vsrlab.vsr.models.VRT.modules import spynet as ref_spy
    net = ref_spy.SpyNet(pretrained=False, return_levels=[2, 3, 4, 5])
    with torch.no_grad():
        for k, p in net.named_parameters():
            p.copy_(keyed_tensor(k, tuple(p.shape)) * (3.0 if k.endswith("weight") else 1.0))    # x3: flows of a few pixels
    store = {"keys": np.asarray(sorted(net.state_dict().keys()))}
    for tag, shape in (("a", (1, 3, 64, 96)), ("b", (2, 3, 40, 72))):
        ref, supp = rand(80 + ord(tag), *shape), rand(90 + ord(tag), *shape)
        with torch.no_grad():
            flows = net(ref, supp)
        for i, f in enumerate(flows):
            store[f"{tag}__flow{i}"] = f.float().numpy()
        store[f"{tag}__seed_ref"], store[f"{tag}__seed_supp"] = np.asarray(80 + ord(tag)), np.asarray(90 + ord(tag))
    np.savez_compressed(os.path.join(HERE, "vrt_spynet.npz"), **store)
    print("vrt_spynet", {k: v.shape for k, v in store.items()})


def vrt_helpers():
    """Deterministic index / buffer helpers of window_attention.py (:9-77, :164-188) as data: relative position index, sine position
    encoding, shift masks (incl. zero shifts along an axis), get_window_size results."""
    import_reference()
    from vsrlab.vsr.models.VRT.modules import window_attention as R
    store = {}
    for name, ws in (("a", (2, 8, 8)), ("b", (6, 8, 8)), ("c", (3, 5, 7))):
        m = R.WindowAttention(120, ws, 6, True, None, True)
        store[f"{name}__ws"] = np.asarray(ws)
        store[f"{name}__index"] = m.relative_position_index.numpy().astype(np.int32)
        store[f"{name}__sine"] = m.position_bias.numpy()
    for i, (D, H, W, ws, ss) in enumerate(((4, 24, 16, (2, 8, 8), (1, 4, 4)), (6, 16, 16, (6, 8, 8), (0, 4, 4)), (4, 16, 16, (2, 8, 8), (1, 0, 4)),
                                           (2, 8, 8, (2, 8, 8), (0, 0, 0)))):
        store[f"m{i}__args"] = np.asarray((D, H, W) + ws + ss)
        store[f"m{i}__mask"] = R.compute_mask(D, H, W, ws, ss, "cpu").numpy().astype(np.int8)
    for i, (xs, ws, ss) in enumerate((((4, 20, 16), (2, 8, 8), (1, 4, 4)), ((6, 16, 16), (6, 8, 8), (3, 4, 4)), ((2, 8, 9), (2, 8, 8), (1, 4, 4)))):
        u, v = R.get_window_size(xs, ws, ss)
        store[f"w{i}__args"] = np.asarray(xs + ws + ss)
        store[f"w{i}__out"] = np.asarray(tuple(u) + tuple(v))
    np.savez_compressed(os.path.join(HERE, "vrt_helpers.npz"), **store)
    print("vrt_helpers", len(store))


def vrt_groups():
    """TMSAG and RTMSA (vsr/models/VRT/modules/tmsa.py:126-251), fp64 reference: a depth-3 TMSAG with mutual attention (dim 120, 6 heads,
    window (2,8,8), volume (4,20,16): padding in H, blocks 0 / 2 unshifted, block 1 shifted by (1,4,4)) and a depth-2 RTMSA (dim 180,
    6 heads, window (6,8,8) on a (6,16,16) volume: the D extent equals the window, so get_window_size zeroes that shift).  Keyed parameters."""
    torch.set_num_threads(8)
    import_reference()
    from vsrlab.vsr.models.VRT.modules import tmsa as ref_tmsa
    from oracle.basicvsr_oracle import keyed_tensor
    store = {}

    def run(tag, m, shape, seeds):
        with torch.no_grad():
            for k, p in m.named_parameters():
                p.copy_(keyed_tensor(k, tuple(p.shape)))
        m = m.double()
        x = rand(seeds[0], *shape, lo=-1, hi=1).double().requires_grad_(True)
        cot = rand(seeds[1], *shape, lo=-1, hi=1).double()
        y = m(x)
        (y * cot).sum().backward()
        store[f"{tag}__out"] = y.detach().float().numpy()
        gx = x.grad.detach()
        store[f"{tag}__gnorm__dx"] = np.asarray(float(gx.norm()))
        store[f"{tag}__gproj__dx"] = np.asarray(float((gx * proj_vector("dx", tuple(gx.shape))).sum()))
        store[f"{tag}__seed_x"], store[f"{tag}__seed_cot"] = np.asarray(seeds[0]), np.asarray(seeds[1])
        store[f"{tag}__keys"] = np.asarray(sorted(m.state_dict().keys()))
        for k, p in m.named_parameters():
            g = p.grad.detach()
            name = k.replace(".", "__")
            store[f"{tag}__gnorm__{name}"] = np.asarray(float(g.norm()))
            store[f"{tag}__gproj__{name}"] = np.asarray(float((g * proj_vector(k, tuple(g.shape))).sum()))

    run("g", ref_tmsa.TMSAG(120, (4, 20, 16), 3, 6, window_size=[2, 8, 8], mut_attn=True, mlp_ratio=2., qkv_bias=True), (1, 120, 4, 20, 16), (91, 92))
    run("r", ref_tmsa.RTMSA(180, (6, 16, 16), 2, 6, window_size=[6, 8, 8], mlp_ratio=2., qkv_bias=True), (1, 180, 6, 16, 16), (93, 94))
    np.savez_compressed(os.path.join(HERE, "vrt_groups.npz"), **store)
    print("vrt_groups", len(store))


def schema():
    """state_dict keys, shapes and requires_grad flags of the reference's own modules at the BASELINE configurations, as JSON
    (data only): BasicVSR(64, 30, 4) (configs[1]), RealBasicVSR(20 cleaning blocks, mid 64, 20 residual blocks) (configs[2]),
    UNetDiscriminator(3, 64), VRT's SpyNet, optical_flow's SpyNet(k=6).  tests/test_host_logic.py compares the
    parameter containers of vsrlab_amd with this file directly."""
    import importlib
    import json
    basicvsr, spynet, realbasicvsr, conv, upsampling = import_reference()
    out = {}

    def dump(name, m):
        req = {k: bool(p.requires_grad) for k, p in m.named_parameters()}
        out[name] = {k: {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""), "requires_grad": req.get(k)}
                     for k, v in m.state_dict().items()}

    dump("BasicVSR(64,30,4,False,False)", basicvsr.BasicVSR(64, 30, 4, False, False))
    dump("RealBasicVSR(20,mid_channels=64,upscale=4,res_blocks=20)", realbasicvsr.RealBasicVSR(20, mid_channels=64, upscale=4, res_blocks=20,
                                                                                                     pretrained_flow=False, train_flow=False))
    mod = importlib.import_module("vsrlab.vsr.models.RealBasicVSR.modules.unet-discriminator")
    dump("UNetDiscriminator(3,64)", mod.UNetDiscriminator(3, 64))
    from vsrlab.vsr.models.VRT.modules import spynet as vrt_spy
    dump("VRT.SpyNet(pretrained=False)", vrt_spy.SpyNet(pretrained=False))
    from vsrlab.vsr.models.VRT.modules import tmsa as ref_tmsa
    dump("TMSAG(120,(4,20,16),3,6,[2,8,8])", ref_tmsa.TMSAG(120, (4, 20, 16), 3, 6, window_size=[2, 8, 8], mut_attn=True, mlp_ratio=2., qkv_bias=True))
    dump("RTMSA(180,(6,16,16),2,6,[6,8,8])", ref_tmsa.RTMSA(180, (6, 16, 16), 2, 6, window_size=[6, 8, 8], mlp_ratio=2., qkv_bias=True))
    with open(os.path.join(HERE, "state_dict_schema.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"), sort_keys=True)
    print("state_dict_schema.json:", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "schema"):
        schema()
    if which in ("all", "vrt"):
        vrt()
    if which in ("all", "vrt_helpers"):
        vrt_helpers()
    if which in ("all", "vrt_groups"):
        vrt_groups()
    if which in ("all", "disc"):
        disc()
    if which in ("all", "realtrain"):
        realtrain()
    if which in ("all", "main"):
        main()
    if which in ("all", "trainflow"):
        trainflow()
    if which in ("all", "lrgrad"):
        lrgrad()
    if which in ("all", "up2"):
        up2()

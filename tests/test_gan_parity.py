"""GPU parity tests of the GAN side (BASELINE config 3): UNetDiscriminator forward / backward,
spectral norm, BCE-with-logits and the two losses of one GAN iteration, through the C ABI, against the reference's own
fp64 outputs (tests/golden/unet_discriminator.npz, gan_step.npz) and the CPU oracle (oracle/discriminator_oracle.py).

Tolerances: fp32 build 1e-3 relative (north-star bar); bf16 build: error <= 1.5 x the error of the oracle that rounds to
bf16 at the same storage points (the noise-floor criterion of test_hip_parity.py)."""
import os

import pytest
import torch
import torch.nn.functional as F

from helpers import golden, proj_vector, rand, rel_err, rel_l2

pytestmark = pytest.mark.gpu

from oracle import basicvsr_oracle as O  # noqa: E402  (checker only)
from oracle import discriminator_oracle as D  # noqa: E402


def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests need an MI355X")
    return torch.device("cuda:0")


def _make_d(dev, dtype):
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.unet_discriminator import UNetDiscriminator
    d = UNetDiscriminator(3, 64)
    d.load_state_dict(O.keyed_state_dict(D.disc_param_shapes(3, 64)), strict=True)
    d = d.to(dev).train()
    d.compute_dtype = dtype
    return d


def test_spectral_norm_kernel_forward_backward():
    dev = _gpu()
    import vsrlab_amd
    from vsrlab_amd import functional as VF
    lib = vsrlab_amd._lib.load()
    for (co, ci, ks) in [(128, 64, 4), (512, 256, 4), (64, 128, 3)]:
        g = torch.Generator().manual_seed(co + ci)
        w = torch.randn(co, ci, ks, ks, generator=g) * 0.05
        u = F.normalize(torch.randn(co, generator=g), dim=0)
        v = F.normalize(torch.randn(ci * ks * ks, generator=g), dim=0)
        w64 = w.double().requires_grad_(True)
        wn, u1, v1, sigma = D.spectral_normalize(w64, u.double(), v.double(), True)
        cot = torch.randn(co, ci, ks, ks, generator=g).double()
        (wn * cot).sum().backward()
        for training in (1, 0):
            wd, ud, vd = w.to(dev), u.clone().to(dev), v.clone().to(dev)
            out, sg = torch.empty_like(wd), torch.empty(1, device=dev)
            st = VF._stream()
            assert lib.vsr_spectral_norm(VF._ptr(wd), VF._ptr(ud), VF._ptr(vd), VF._ptr(out), VF._ptr(sg), co, ci * ks * ks, training, st) == 0
            if training:
                assert rel_err(ud.cpu(), u1) < 1e-5 and rel_err(vd.cpu(), v1) < 1e-5
                assert rel_err(out.cpu(), wn.detach()) < 1e-5 and abs(float(sg) - float(sigma)) < 1e-5 * float(sigma)
                gd = torch.zeros_like(wd)
                scr = torch.empty(1024, device=dev)
                assert lib.vsr_spectral_norm_backward(VF._ptr(cot.float().to(dev)), VF._ptr(wd), VF._ptr(ud), VF._ptr(vd), VF._ptr(sg),
                                                      VF._ptr(gd), co, ci * ks * ks, VF._ptr(scr), st) == 0
                assert rel_err(gd.cpu(), w64.grad) < 1e-4
            else:
                wn0, _, _, _ = D.spectral_normalize(w.double(), u.double(), v.double(), False)
                assert torch.equal(ud.cpu(), u) and torch.equal(vd.cpu(), v)           # eval: buffers untouched
                assert rel_err(out.cpu(), wn0) < 1e-5


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_spectral_conv_standalone_vs_torch(dtype):
    """SpectralConv(64, 64) on its own (core/modules/conv.py:6-13; round-3 VERDICT missing #3) against the reference's definition
    itself -- torch.nn.utils.spectral_norm(nn.Conv2d(64, 64, 3, 1, 1, bias=False)) on the CPU in float64 -- over two training-mode
    forwards (the buffers move) and one eval forward: output, d/dx, d/d weight_orig, weight_u / weight_v."""
    dev = _gpu()
    import torch.nn as nn
    from torch.nn.utils import spectral_norm
    from vsrlab_amd.core.modules.conv import SpectralConv
    torch.manual_seed(3)
    ref = spectral_norm(nn.Conv2d(64, 64, 3, 1, 1, bias=False)).double()
    m = SpectralConv(64, 64)
    with torch.no_grad():
        m.conv.weight_orig.copy_(ref.weight_orig.float())
        m.conv.weight_u.copy_(ref.weight_u.float())
        m.conv.weight_v.copy_(ref.weight_v.float())
    m = m.to(dev)
    os.environ["VSRLAB_AMD_DTYPE"] = dtype
    t_out, t_g = (1e-3, 1e-3) if dtype == "fp32" else (2e-2, 3e-2)
    try:
        for step, training in enumerate((True, True, False)):
            ref.train(training)
            m.train(training)
            x = rand(60 + step, 2, 64, 21, 40, lo=-1, hi=1)
            cot = rand(70 + step, 2, 64, 21, 40, lo=-1, hi=1)
            xr = x.double().requires_grad_(True)
            ref.weight_orig.grad = None
            yr = ref(xr)                                     # ONE call per step: a training-mode forward moves the buffers
            (yr * cot.double()).sum().backward()
            xd = x.to(dev).requires_grad_(True)
            m.conv.weight_orig.grad = None
            y = m(xd)
            (y * cot.to(dev)).sum().backward()
            assert rel_err(y, yr) < t_out, (step, rel_err(y, yr))
            assert rel_l2(xd.grad, xr.grad) < t_g and rel_l2(m.conv.weight_orig.grad, ref.weight_orig.grad) < t_g, step
            assert rel_err(m.conv.weight_u, ref.weight_u) < 1e-4 and rel_err(m.conv.weight_v, ref.weight_v) < 1e-4, step
    finally:
        del os.environ["VSRLAB_AMD_DTYPE"]
    with pytest.raises(NotImplementedError):
        SpectralConv(64, 128, 4, 2, 1).to(dev)(rand(1, 1, 64, 16, 16).to(dev))


def test_bce_with_logits_and_adversarial_loss():
    dev = _gpu()
    from vsrlab_amd.core.losses import AdversarialLoss
    x = (rand(5, 2, 1, 24, 40) * 8 - 4).requires_grad_(True)
    adv = AdversarialLoss()
    for target, is_disc in ((1, False), (1, True), (0, True)):
        want = D.adversarial_loss(x, target, is_disc)
        (gw,) = torch.autograd.grad(want, x)
        xd = x.detach().to(dev).requires_grad_(True)
        got = adv(xd, target, is_disc)
        got.backward()
        assert abs(float(got) - float(want)) < 1e-5 * abs(float(want))
        assert rel_err(xd.grad.cpu(), gw) < 1e-5


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_unet_discriminator_vs_golden(dtype):
    """(2,3,32,48) frames: logits, d/d img, all 12 parameter gradients and the spectral-norm buffers after the
    training-mode forward, against the reference's fp64 run."""
    dev = _gpu()
    g = golden("unet_discriminator")
    d = _make_d(dev, dtype)
    img = rand(int(g["seed_img"]), 2, 3, 32, 48).to(dev).requires_grad_(True)
    cot = rand(int(g["seed_cot"]), 2, 1, 32, 48, lo=-1, hi=1).to(dev)
    out = d(img)
    torch.mean(out * cot).backward()
    sd = {k: v.double() for k, v in O.keyed_state_dict(D.disc_param_shapes(3, 64)).items()}
    # fp64 oracle (pinned to the golden by tests/test_oracle_golden.py): full gradients for every tensor
    leaves = {k: v.clone().requires_grad_(not k.endswith(("weight_u", "weight_v"))) for k, v in sd.items()}
    img64 = rand(int(g["seed_img"]), 2, 3, 32, 48).double().requires_grad_(True)
    out64 = D.discriminator_forward(leaves, img64, True)
    torch.mean(out64 * cot.cpu().double()).backward()
    assert rel_err(out64, g["out"]) < 1e-6
    named = dict(d.named_parameters())
    ref = {k: v.grad for k, v in leaves.items() if v.requires_grad}
    got = {k: named[k].grad.detach().cpu() for k in ref}
    got["__dimg"], ref["__dimg"] = img.grad.detach().cpu(), img64.grad
    if dtype == "fp32":
        assert rel_err(out, g["out"]) < 1e-3 and rel_err(img.grad, g["dimg"]) < 1e-3
        for k in ref:
            assert rel_l2(got[k], ref[k]) < 2e-3, (k, rel_l2(got[k], ref[k]))
        for k, v in d.state_dict().items():
            if k.endswith(("weight_u", "weight_v")):
                assert rel_err(v.cpu(), g["buf__" + k.replace(".", "__")]) < 1e-4, k
        for k in named:                                         # the reference's own numbers, tensor by tensor
            tag = k.replace(".", "__")
            gr = named[k].grad.detach().cpu().double()
            assert abs(float(gr.norm()) - float(g["gnorm__" + tag])) < 2e-3 * float(g["gnorm__" + tag]), k
            assert abs(float((gr * proj_vector(k, tuple(gr.shape))).sum()) - float(g["gproj__" + tag])) < 2e-3 * float(g["gnorm__" + tag]) * gr.numel() ** 0.5, k
    else:
        sd32 = {k: v.float() for k, v in sd.items()}
        lv = {k: v.clone().requires_grad_(not k.endswith(("weight_u", "weight_v"))) for k, v in sd32.items()}
        img_e = rand(int(g["seed_img"]), 2, 3, 32, 48).requires_grad_(True)
        with O.emulate_bf16():
            out_e = D.discriminator_forward(lv, img_e, True)
            torch.mean(out_e * cot.cpu()).backward()
        emu = {k: v.grad for k, v in lv.items() if v.requires_grad}
        emu["__dimg"] = img_e.grad
        assert rel_err(out, g["out"]) <= 1.5 * max(rel_err(out_e, g["out"]), 1e-3)
        from test_hip_parity import _noise_floor_check
        _noise_floor_check(got, emu, ref, max_glob_ratio=1.5, max_tensor_ratio=2.5)


def test_discriminator_bf16_many_tiles_per_workgroup():
    """conv_wide2 is persistent: at config 3 a workgroup walks ~125 tiles, at the golden's 32x48 frames exactly one.  Here the
    workgroups per output column are capped at 3 (VSRLAB_AMD_WIDE2_MAX_WG, a test knob), so on a (1,3,72,136) frame every wide layer
    runs several tiles per workgroup (ragged ones included): tile-to-tile DMA ring, epilogue drain, K-split reduction.  bf16 vs the
    fp64 oracle with the noise-floor criterion."""
    dev = _gpu()
    import os
    from test_hip_parity import _noise_floor_check
    os.environ["VSRLAB_AMD_WIDE2_MAX_WG"] = "3"
    try:
        d = _make_d(dev, "bf16")
        img = rand(21, 1, 3, 72, 136).to(dev).requires_grad_(True)
        cot = rand(22, 1, 1, 72, 136, lo=-1, hi=1).to(dev)
        out = d(img)
        torch.mean(out * cot).backward()
    finally:
        del os.environ["VSRLAB_AMD_WIDE2_MAX_WG"]
    sd = {k: v.double() for k, v in O.keyed_state_dict(D.disc_param_shapes(3, 64)).items()}
    leaves = {k: v.clone().requires_grad_(not k.endswith(("weight_u", "weight_v"))) for k, v in sd.items()}
    img64 = rand(21, 1, 3, 72, 136).double().requires_grad_(True)
    out64 = D.discriminator_forward(leaves, img64, True)
    torch.mean(out64 * cot.cpu().double()).backward()
    named = dict(d.named_parameters())
    ref = {k: v.grad for k, v in leaves.items() if v.requires_grad}
    got = {k: named[k].grad.detach().cpu() for k in ref}
    got["__dimg"], ref["__dimg"] = img.grad.detach().cpu(), img64.grad
    lv = {k: v.float().clone().requires_grad_(not k.endswith(("weight_u", "weight_v"))) for k, v in sd.items()}
    img_e = rand(21, 1, 3, 72, 136).requires_grad_(True)
    with O.emulate_bf16():
        out_e = D.discriminator_forward(lv, img_e, True)
        torch.mean(out_e * cot.cpu()).backward()
    emu = {k: v.grad for k, v in lv.items() if v.requires_grad}
    emu["__dimg"] = img_e.grad
    assert rel_err(out, out64) <= 1.5 * max(rel_err(out_e, out64), 1e-3), (rel_err(out, out64), rel_err(out_e, out64))
    _noise_floor_check(got, emu, ref, max_glob_ratio=1.5, max_tensor_ratio=2.5)


def test_gan_iteration_vs_golden():
    """generator_step / discriminator_step (train_gan.py:35-58, perceptual_loss null) on fixed sr / lq / hr tensors: both
    losses, d loss_g / d sr, d loss_g / d lq, D's parameter gradients of loss_d and the u / v buffers after the three
    training-mode forwards, against the reference-generated golden (fp32 build)."""
    dev = _gpu()
    from vsrlab_amd.core.losses import AdversarialLoss, CharbonnierLoss
    from vsrlab_amd.core.utils import compute_loss
    g = golden("gan_step")
    d = _make_d(dev, "fp32")
    b, t, c, h, w = 1, 2, 3, 32, 48
    sr = rand(int(g["seed_sr"]), b, t, c, h, w).to(dev).requires_grad_(True)
    hr = rand(int(g["seed_hr"]), b, t, c, h, w).to(dev)
    lq = rand(int(g["seed_lq"]), b, t, c, h // 4, w // 4).to(dev).requires_grad_(True)
    adv, crit = AdversarialLoss(), CharbonnierLoss()
    # generator side (generator_step with the model's outputs given)
    loss_g = compute_loss(crit, sr, hr, lq) + 0.0 + adv(d(sr.reshape(-1, c, h, w)), 1, False)
    loss_g.backward()
    assert abs(float(loss_g) - float(g["loss_g"])) < 1e-5 * float(g["loss_g"])
    assert rel_err(sr.grad, g["dsr"]) < 1e-3 and rel_err(lq.grad, g["dlq"]) < 1e-3
    d.zero_grad()
    from vsrlab_amd.train_gan import discriminator_step
    loss_d = discriminator_step(d, adv, sr, hr)
    loss_d.backward()
    assert abs(float(loss_d) - float(g["loss_d"])) < 1e-5 * float(g["loss_d"])
    for k, p in d.named_parameters():
        tag = k.replace(".", "__")
        gr = p.grad.detach().cpu().double()
        assert abs(float(gr.norm()) - float(g["gnorm__" + tag])) < 2e-3 * float(g["gnorm__" + tag]), k
        assert abs(float((gr * proj_vector(k, tuple(gr.shape))).sum()) - float(g["gproj__" + tag])) < 2e-3 * float(g["gnorm__" + tag]) * gr.numel() ** 0.5, k
    for k, v in d.state_dict().items():
        if k.endswith(("weight_u", "weight_v")):
            assert rel_err(v.cpu(), g["buf__" + k.replace(".", "__")]) < 1e-4, k


def test_discriminator_larger_frame_and_eval_mode():
    """(1,3,72,136): several tiles per level incl. ragged ones (136/8 = 17 columns at H/8); eval mode leaves u / v untouched;
    a second backward through the same graph raises."""
    dev = _gpu()
    d = _make_d(dev, "fp32")
    sd = O.keyed_state_dict(D.disc_param_shapes(3, 64))
    img = rand(9, 1, 3, 72, 136)
    want = D.discriminator_forward(sd, img, True)
    x = img.to(dev).requires_grad_(True)
    out = d(x)
    assert rel_err(out, want) < 1e-3
    loss = out.mean()
    loss.backward()
    with pytest.raises(RuntimeError):
        loss.backward()
    d.eval()
    before = {k: v.clone() for k, v in d.state_dict().items() if k.endswith(("weight_u", "weight_v"))}
    with torch.no_grad():
        out_e = d(img.to(dev))
    after = d.state_dict()
    assert all(torch.equal(after[k], v) for k, v in before.items())
    sd_after = {k: v.cpu() for k, v in after.items()}
    assert rel_err(out_e, D.discriminator_forward(sd_after, img, False)) < 1e-3
    with pytest.raises(RuntimeError):
        d(torch.zeros(1, 3, 30, 48, device=dev))                 # height not a multiple of 8: loud, no fallback


def test_config3_full_size_gan_iteration():
    """BASELINE config 3 at its frame size, one GAN iteration through the reference-shaped step functions:
    RealBasicVSR(cleaning_blocks=20, mid 64, res_blocks=20: conf/train/model/basicvsr.yaml) on a (1,7,3,540,960) clip, D on the
    seven 2160x3840 frames, bf16.  generator_step -> backward -> FusedAdam, discriminator_step -> backward -> FusedAdam: finite
    losses and gradients, parameters move, a timing line.  (Values are pinned at small sizes above; this is the at-size run:
    ~230 GB of arenas.)"""
    dev = _gpu()
    import gc
    import time
    gc.collect()                      # earlier tests' autograd graphs (reference cycles) may still hold their arenas: this test needs 241 GiB
    torch.cuda.empty_cache()
    from vsrlab_amd.core.losses import AdversarialLoss, CharbonnierLoss
    from vsrlab_amd.optim import FusedAdam
    from vsrlab_amd.train_gan import discriminator_step, dummy_loss, generator_step
    from vsrlab_amd.vsr.models.RealBasicVSR.realbasicvsr import RealBasicVSR
    torch.manual_seed(0)
    g = RealBasicVSR(20, mid_channels=64, upscale=4, res_blocks=20, pretrained_flow=False, train_flow=False).to(dev)
    g.basicvsr.compute_dtype = "bf16"
    os_env = __import__("os").environ
    os_env["VSRLAB_AMD_DTYPE"] = "bf16"
    try:
        d = _make_d(dev, "bf16")
        opt_g = FusedAdam(g.parameters(), lr=1e-4, betas=(0.9, 0.99))
        opt_d = FusedAdam(d.parameters(), lr=1e-4, betas=(0.9, 0.99))
        lr = torch.rand(1, 7, 3, 540, 960, device=dev)
        hr = torch.rand(1, 7, 3, 2160, 3840, device=dev)
        adv, crit = AdversarialLoss(), CharbonnierLoss()
        w_g0, w_d0 = opt_g.flat_params.clone(), opt_d.flat_params.clone()
        times = []
        for it in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sr, loss_g, perceptual_g, adv_g = generator_step(g, d, crit, dummy_loss, adv, lr, hr)
            loss_g.backward()
            opt_g.step(max_grad_norm=1.0)
            opt_g.zero_grad()
            loss_d = discriminator_step(d, adv, sr, hr)
            loss_d.backward()
            opt_d.step(max_grad_norm=1.0)
            opt_d.zero_grad()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            assert bool(torch.isfinite(loss_g)) and bool(torch.isfinite(loss_d)), (float(loss_g), float(loss_d))
        assert bool(torch.isfinite(opt_g.flat_params).all()) and bool(torch.isfinite(opt_d.flat_params).all())
        assert float((opt_g.flat_params - w_g0).abs().max()) > 0 and float((opt_d.flat_params - w_d0).abs().max()) > 0
        assert float(opt_g.last_grad_norm) > 0 and float(opt_d.last_grad_norm) > 0
        print(f"config 3 (RealBasicVSR(20 clean, 20 res) + UNetDiscriminator, 540p x7, bf16): GAN iteration {times[-1] * 1e3:.0f} ms "
              f"(first {times[0] * 1e3:.0f} ms), loss_g {float(loss_g):.4f}, loss_d {float(loss_d):.4f}, "
              f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.0f} GiB")
    finally:
        del os_env["VSRLAB_AMD_DTYPE"]


# ---- round-2 ADVICE: the GAN step functions open the reference's autocast regions (train_gan.py:38-42,52-56) ----
@pytest.mark.gpu
def test_gan_steps_open_the_references_autocast_regions():
    _gpu()                                                 # torch disables a "cuda" autocast region on a box without a GPU
    from vsrlab_amd import functional as VF
    from vsrlab_amd import train_gan as TG
    seen = []

    def model(lr):                                         # stands for RealBasicVSR: records what its forward would resolve
        seen.append(("model", VF.resolve_dtype()))
        return torch.zeros(1, 2, 3, 8, 8, requires_grad=True), None

    def disc(x):
        seen.append(("disc", VF.resolve_dtype()))
        return x.mean(dim=(1, 2, 3))

    import os
    os.environ.pop("VSRLAB_AMD_DTYPE", None)
    assert VF.resolve_dtype() == VF.DT_F32                # outside the step functions nothing is overridden
    hr = torch.zeros(1, 2, 3, 8, 8)
    sr, loss, _, _ = TG.generator_step(model, disc, lambda a, b: (a - b).abs().mean(), TG.dummy_loss,
                                       lambda logits, target, is_disc: logits.mean(), torch.zeros(1, 2, 3, 2, 2), hr)
    TG.discriminator_step(disc, lambda logits, target, is_disc: logits.mean(), sr, hr)
    assert [k for k, _ in seen] == ["model", "disc", "disc", "disc"]
    assert all(d == VF.DT_BF16 for _, d in seen), seen    # the bf16 build without compute_dtype / $VSRLAB_AMD_DTYPE
    assert VF.resolve_dtype() == VF.DT_F32

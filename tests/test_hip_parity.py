"""GPU parity tests: every call goes through the C ABI (libvsrlab_hip.so via vsrlab_amd) and is
checked against the CPU oracle (oracle/basicvsr_oracle.py, itself pinned to the reference by
tests/golden) and against the committed golden vectors.

Tolerances
  fp32 build : 1e-3 relative (max|a-b|/max|b|) -- the north-star bar; observed ~1e-5 forward.
               Weight gradients are additionally bounded in relative L2 because ReLU/LeakyReLU
               masks are discontinuous: two exact-fp32 implementations differ by up to ~2e-3 in
               max-norm on a few elements (measured: reference fp32 vs reference fp64).
  bf16 build : inputs/weights are rounded to bf16 and every stored activation is bf16 (8 bits of
               mantissa), so: per-op 1e-2 relative to the oracle run on the SAME bf16-rounded
               inputs; end-to-end 4e-2 (sr) / 8e-2 relative L2 (grads) vs the fp32 oracle.
"""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

from helpers import golden, rand, rel_err, rel_l2

pytestmark = pytest.mark.gpu

from oracle import basicvsr_oracle as O  # noqa: E402  (checker only)


def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests need an MI355X")
    return torch.device("cuda:0")


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def tol(dtype, fp32, bf16):
    return fp32 if dtype == "fp32" else bf16


DTYPES = ["fp32", "bf16"]


# --------------------------------------------------------------------------------------------- #
def test_library_loaded_and_abi():
    import vsrlab_amd
    lib = vsrlab_amd._lib.load()
    assert lib.vsr_abi_version() == 4
    _gpu()


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_roundtrip(dtype):
    from vsrlab_amd import functional as VF
    dev = _gpu()
    x = rand(21, 2, 5, 9, 11, lo=-1, hi=1)
    pm = VF.to_pixel_major(x.to(dev), VF.resolve_dtype(dtype))
    assert pm.shape == (2, 9, 1, 2, 32, 8) and pm.pm_w == 11      # blocked: [N][H][W/32][C/8][32][8]
    back = VF.from_pixel_major(pm, 5).cpu()
    ref = x if dtype == "fp32" else bf16_round(x)
    assert torch.equal(back, ref)
    assert float(VF.from_pixel_major(pm).cpu()[:, 5:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_flow_warp_forward_backward_vs_oracle_and_golden(dtype):
    from vsrlab_amd import functional as VF
    dev = _gpu()
    g = golden("flow_warp")
    x = rand(g["seed_x"], 2, 5, 9, 11, lo=-1, hi=1)
    flow = rand(g["seed_flow"], 2, 2, 9, 11, lo=-8, hi=8)
    xin = x if dtype == "fp32" else bf16_round(x)
    xg = xin.clone().to(dev).requires_grad_(True)
    out = VF.flow_warp(xg, flow.permute(0, 2, 3, 1).to(dev), compute_dtype=dtype)
    cot = rand(22, 2, 5, 9, 11, lo=-1, hi=1)
    cot_in = cot if dtype == "fp32" else bf16_round(cot)
    out.backward(cot_in.to(dev))
    xo = xin.clone().requires_grad_(True)
    ref = O.flow_warp(xo, flow, "zeros")
    ref.backward(cot_in)
    assert rel_err(out, ref) < tol(dtype, 1e-5, 1e-2)
    assert rel_err(xg.grad, xo.grad) < tol(dtype, 1e-5, 1e-5)      # scatter accumulates in fp32 in both builds
    if dtype == "fp32":
        assert rel_err(out, g["zeros"]) < 1e-5                      # the reference's own output


def test_flow_warp_border_mode_vs_oracle_and_golden():
    """flow_warp(x, flow, padding_mode='border') (spynet.py:60,95): the reference's own output (golden), and the oracle's
    gradients w.r.t. x and w.r.t. the flow (clamped coordinates carry no flow gradient), fp32 build."""
    from vsrlab_amd import functional as VF
    dev = _gpu()
    g = golden("flow_warp")
    x = rand(g["seed_x"], 2, 5, 9, 11, lo=-1, hi=1)
    flow = rand(g["seed_flow"], 2, 2, 9, 11, lo=-8, hi=8)
    cot = rand(23, 2, 5, 9, 11, lo=-1, hi=1)
    xg = x.clone().to(dev).requires_grad_(True)
    fg = flow.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    out = VF.flow_warp(xg, fg, padding_mode="border", compute_dtype="fp32")
    out.backward(cot.to(dev))
    xo, fo = x.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    ref = O.flow_warp(xo, fo, "border")
    ref.backward(cot)
    assert rel_err(out, g["border"]) < 1e-5 and rel_err(out, ref) < 1e-5
    assert rel_err(xg.grad, xo.grad) < 1e-5
    assert rel_err(fg.grad.permute(0, 3, 1, 2), fo.grad) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(1, 8, 32), (2, 13, 37), (1, 40, 72)])
def test_conv3x3_c64_fwd_dgrad_wgrad(dtype, shape):
    """The trunk's hot kernel family (conv.py:85-86) incl. ragged tiles (H,W not multiples of 8,32)."""
    from vsrlab_amd import functional as VF
    dev = _gpu()
    n, h, w = shape
    dt = VF.resolve_dtype(dtype)
    rnd = (lambda t: t) if dtype == "fp32" else bf16_round
    x = rnd(rand(31, n, 64, h, w, lo=-1, hi=1))
    wgt = rnd(O.keyed_tensor("test.conv.weight", (64, 64, 3, 3)))
    b = O.keyed_tensor("test.conv.bias", (64,))
    res = rnd(rand(32, n, 64, h, w, lo=-1, hi=1))
    dy = rnd(rand(33, n, 64, h, w, lo=-1, hi=1))
    xp, rp, dyp = (VF.to_pixel_major(t.to(dev), dt) for t in (x, res, dy))
    t_out = tol(dtype, 2e-5, 1e-2)
    # forward: relu(conv + b) + res
    y = VF.from_pixel_major(VF.conv3x3_c64(xp, wgt.to(dev), b.to(dev), act=1, res_pm=rp)).cpu()
    ref = F.relu(F.conv2d(x, wgt, b, padding=1)) + res
    assert rel_err(y, ref) < t_out
    # leaky variant, no residual
    y = VF.from_pixel_major(VF.conv3x3_c64(xp, wgt.to(dev), b.to(dev), act=2)).cpu()
    assert rel_err(y, F.leaky_relu(F.conv2d(x, wgt, b, padding=1), 0.1)) < t_out
    # data gradient with ReLU mask and with residual
    dref = F.conv_transpose2d(dy, wgt, padding=1)
    dx = VF.from_pixel_major(VF.conv3x3_c64_dgrad(dyp, wgt.to(dev), aux_pm=rp, mask_mode=1)).cpu()
    assert rel_err(dx, dref * (res > 0)) < t_out
    dx = VF.from_pixel_major(VF.conv3x3_c64_dgrad(dyp, wgt.to(dev), res_pm=rp, aux_pm=xp, mask_mode=2)).cpu()
    assert rel_err(dx, (dref + res) * torch.where(x > 0, 1.0, 0.1)) < t_out
    # weight / bias gradient (fp32 accumulation, fp32 output in both builds)
    gw, gb = VF.conv3x3_c64_wgrad(xp, dyp)
    xr = x.clone().requires_grad_(False)
    wr = wgt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    (F.conv2d(xr, wr, br, padding=1) * dy).sum().backward()
    assert rel_err(gw.cpu(), wr.grad) < 2e-5
    assert rel_err(gb.cpu(), br.grad) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_residual_conv_module_fwd_bwd(dtype):
    from vsrlab_amd.core.modules.conv import ResidualConv
    dev = _gpu()
    sd = O.keyed_state_dict({f"conv{j}.{p}": s for j in (1, 2) for p, s in (("weight", (64, 64, 3, 3)), ("bias", (64,)))})
    m = ResidualConv(64)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    x = rand(41, 2, 64, 20, 36, lo=-1, hi=1)
    cot = rand(42, 2, 64, 20, 36, lo=-1, hi=1)
    import os
    os.environ["VSRLAB_AMD_DTYPE"] = dtype
    try:
        xg = x.clone().to(dev).requires_grad_(True)
        y = m(xg)
        y.backward(cot.to(dev))
    finally:
        del os.environ["VSRLAB_AMD_DTYPE"]
    leaves = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
    xo = x.clone().double().requires_grad_(True)
    yo = O.residual_conv(leaves, "", xo)
    yo.backward(cot.double())
    assert rel_err(y, yo) < tol(dtype, 1e-4, 2e-2)
    assert rel_l2(xg.grad, xo.grad) < tol(dtype, 1e-3, 3e-2)
    for k, p in m.named_parameters():
        assert rel_l2(p.grad, leaves[k].grad) < tol(dtype, 1e-3, 8e-2), k   # bf16: ReLU-mask flips on ~0.3% of elements


@pytest.mark.parametrize("dtype", DTYPES)
def test_spynet_vs_oracle_and_golden(dtype):
    """40x72 exercises the resize-to-/32 path, 32x32 the 1x1 coarsest level."""
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.spynet import Spynet
    dev = _gpu()
    g = golden("spynet")
    m = Spynet(False)
    m.load_state_dict(O.keyed_state_dict(O.spynet_param_shapes()), strict=True)
    m = m.to(dev).eval()
    for p in m.parameters():
        p.requires_grad_(False)
    import os
    os.environ["VSRLAB_AMD_DTYPE"] = dtype
    try:
        f1 = m(rand(g["seed_ref"], 2, 3, 40, 72).to(dev), rand(g["seed_supp"], 2, 3, 40, 72).to(dev)).cpu()
        f2 = m(rand(g["seed_ref2"], 1, 3, 32, 32).to(dev), rand(g["seed_supp2"], 1, 3, 32, 32).to(dev)).cpu()
    finally:
        del os.environ["VSRLAB_AMD_DTYPE"]
    # flows reach ~13 px; bf16: six levels of 5 bf16 convs feeding a x2-per-level recursion
    assert rel_err(f1, g["flow"]) < tol(dtype, 1e-3, 5e-2)
    assert rel_err(f2, g["flow2"]) < tol(dtype, 1e-3, 5e-2)


def _run_basicvsr(dtype, mid, blocks, shape, seed_lr, seed_cot, dev, charbonnier_hr=None, train_flow=False):
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.basicvsr import BasicVSR
    m = BasicVSR(mid, blocks, 4, False, train_flow)
    m.load_state_dict(O.keyed_state_dict(O.basicvsr_param_shapes(mid, blocks, 4)), strict=True)
    m = m.to(dev)
    m.compute_dtype = dtype
    n, t, _, h, w = shape
    lrs = rand(seed_lr, *shape)
    cot = rand(seed_cot, n, t, 3, 4 * h, 4 * w, lo=-1, hi=1)
    sr = m(lrs.to(dev))
    torch.mean(sr * cot.to(dev)).backward()
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
    return m, lrs, cot, sr.detach().cpu(), grads


def _grad_report(grads, ref):
    """(global relative L2 over all tensors, worst per-tensor relative L2, cosine of the full gradient)."""
    keys = sorted(ref)
    g = torch.cat([grads[k].double().flatten() for k in keys])
    r = torch.cat([ref[k].double().flatten() for k in keys])
    worst = max((rel_l2(grads[k], ref[k]), k) for k in keys)
    cos = float(torch.dot(g, r) / (g.norm() * r.norm()))
    return rel_l2(g, r), worst, cos


def _noise_floor_check(grads_hip, grads_emu, ref, max_glob_ratio=1.5, max_tensor_ratio=None, floor=1e-3):
    """bf16 criterion (round-1 VERDICT weak #1): the bf16 build is an evaluation of the net with bf16 STORAGE; the CPU
    oracle under emulate_bf16() is another one that rounds at the same points.  Both are compared with the same
    exact reference `ref` (fp64 oracle or the reference's own fp64 golden); the HIP build's error may not exceed the
    emulation's own error by more than the stated ratio -- globally (all tensors concatenated) and per tensor.  A
    fixed tolerance cannot tell a correct bf16 kernel from one with a 10 % bug; this one scales with the depth of
    the net under test.  Returns (global ratio, worst per-tensor ratio, key)."""
    keys = sorted(ref)
    cat = lambda d: torch.cat([d[k].double().flatten() for k in keys])
    r = cat(ref)
    e_hip, e_emu = rel_l2(cat(grads_hip), r), rel_l2(cat(grads_emu), r)
    per = []
    for k in keys:
        eh, ee = rel_l2(grads_hip[k], ref[k]), rel_l2(grads_emu[k], ref[k])
        per.append((eh / max(ee, floor), k, eh, ee))
    worst = max(per)
    assert e_hip <= max_glob_ratio * max(e_emu, floor), ("global", e_hip, e_emu)
    if max_tensor_ratio is not None:
        assert worst[0] <= max_tensor_ratio, ("per-tensor", worst)
    return e_hip / max(e_emu, floor), worst[0], worst[1]


FP32_FLOOR = 2e-4      # fp32 criterion: per-tensor errors below this (plain summation-order rounding, no mask flip) all count as "at the floor"


def _fp32_noise_floor_check(grads_hip, grads_o32, ref64, max_glob_ratio=1.5, max_tensor_ratio=2.5):
    """fp32 criterion (round-2 VERDICT weak #1), the same construction as the bf16 one: the HIP fp32 build and the fp32 CPU oracle
    are two fp32 evaluations of the net; both are compared with the fp64 oracle, and the HIP build's error may not exceed the
    fp32 oracle's own error by more than the stated ratios.  (ReLU / LeakyReLU mask flips make two correct fp32 evaluations
    differ by up to ~1e-2 on single tensors at depth 61 x 5 frames: a fixed tolerance either hides a 1 % bug or fails a
    correct kernel; this one scales with the net under test.)"""
    return _noise_floor_check(grads_hip, grads_o32, ref64, max_glob_ratio=max_glob_ratio, max_tensor_ratio=max_tensor_ratio, floor=FP32_FLOOR)


# Gradient tolerances.  ReLU / LeakyReLU masks are discontinuous, so a pre-activation perturbed by eps
# flips the mask of a fraction ~eps of the elements and the gradient moves by ~sqrt(eps) in relative L2:
#   fp32 (eps ~ 1e-6): measured on the CPU, reference fp32 vs reference fp64 = 1.7e-3 max-norm on
#       conv_last.0.weight (2 blocks); oracle fp32 vs oracle fp64 at config 1 (30 blocks, t=5) = 4.1e-3 worst
#       per-tensor L2.  HIP fp32 vs oracle: global 1.0e-3, worst per-tensor 8.6e-3 (gpu_diag.py).
#   bf16 (eps ~ 4e-3): ~6 % per layer of depth; HIP bf16 vs the bf16-emulating oracle at config 1:
#       global 8-16 %, worst per-tensor 33 %, sr 2.8e-2 (gpu_diag.py) -- two correct bf16 evaluations of a
#       61-conv x 5-frame recurrence differ by that much from each other; the bf16 kernels themselves are
#       pinned per-op at rounding level in the tests above.
@pytest.mark.parametrize("dtype", DTYPES)
def test_basicvsr_end_to_end_vs_golden(dtype):
    """(2,3,3,24,40), mid 64, 3 blocks: sr + 11 parameter gradients against the reference's own
    (fp64) outputs, and the flows against the reference's compute_flow."""
    dev = _gpu()
    g = golden("basicvsr_m64_rb3")
    shape = (2, 3, 3, 24, 40)
    m, lrs, cot, sr, grads = _run_basicvsr(dtype, 64, 3, shape, int(g["seed_lr"]), int(g["seed_cot"]), dev)
    assert rel_err(sr, g["sr"]) < tol(dtype, 1e-3, 1e-2)
    ref = {k[len("grad__"):].replace("__", "."): v for k, v in g.items() if k.startswith("grad__")}
    assert len(ref) == 11
    glob, worst, cos = _grad_report(grads, ref)
    if dtype == "fp32":
        assert glob < 1e-3, (glob, worst)
        assert worst[0] < 5e-3, worst
        assert cos > 0.999999
    else:
        # noise floor of bf16 storage on THIS net (3 blocks, t=3): the oracle with bf16 rounding at the same points
        sd = O.keyed_state_dict(O.basicvsr_param_shapes(64, 3, 4))
        with O.emulate_bf16():
            sr_e, _, g_e = O.fwd_bwd(sd, lrs, torch.zeros_like(cot), cot=cot)
        assert rel_err(sr, g["sr"]) <= 1.5 * max(rel_err(sr_e, g["sr"]), 1e-3)
        _noise_floor_check(grads, g_e, ref, max_glob_ratio=1.5, max_tensor_ratio=2.0)
        assert cos > 0.995
    assert not any(k.startswith("spynet") for k in grads)
    # flows computed inside the engine (vsr_basicvsr_get_flows on a caller-held workspace) and by the module's compute_flow
    import vsrlab_amd
    from vsrlab_amd import functional as VF
    from vsrlab_amd._order import basicvsr_keys
    lib = vsrlab_amd._lib.load()
    keys, _ = basicvsr_keys(3)
    sd = O.keyed_state_dict(O.basicvsr_param_shapes(64, 3, 4))
    ps = [sd[k].to(dev).contiguous() for k in keys]
    dt = VF.resolve_dtype(dtype)
    desc = vsrlab_amd._lib.BasicVSRDesc(2, 3, 24, 40, 64, 3, 4, dt)
    nbytes = lib.vsr_basicvsr_workspace_bytes(ctypes.byref(desc), 0)
    ws = VF.Workspace(nbytes, dev)
    sr2 = torch.empty(2, 3, 3, 96, 160, device=dev)
    st = VF._stream()
    assert lib.vsr_basicvsr_forward(ctypes.byref(desc), VF._ptr_array(ps), len(ps), VF._ptr(lrs.to(dev)), VF._ptr(sr2), VF._ptr(ws.buf), nbytes, 0, st) == 0
    ff, fb = VF.basicvsr_flows(shape, 64, 3, 4, ws, dt, dev)
    assert rel_err(sr2, g["sr"]) < tol(dtype, 1e-3, 1e-2)               # inference schedule, same values
    assert rel_err(ff.reshape(-1, 2, 24, 40).cpu(), g["flow_forward"]) < tol(dtype, 1e-3, 5e-2)
    assert rel_err(fb.reshape(-1, 2, 24, 40).cpu(), g["flow_backward"]) < tol(dtype, 1e-3, 5e-2)
    with torch.no_grad():
        cf, cb = m.compute_flow(lrs.to(dev))                            # the reference helper (basicvsr.py:30-37)
    assert rel_err(cf.cpu(), g["flow_forward"]) < tol(dtype, 1e-3, 5e-2) and rel_err(cb.cpu(), g["flow_backward"]) < tol(dtype, 1e-3, 5e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_flow_warp_flow_gradient_vs_oracle(dtype):
    """d flow_warp / d flow (grid_sampler_2d_backward's grid gradient, zeros padding), incl. out-of-range taps."""
    from vsrlab_amd import functional as VF
    dev = _gpu()
    x = rand(1, 2, 64, 9, 11, lo=-1, hi=1)
    flow = rand(2, 2, 2, 9, 11, lo=-8, hi=8)
    cot = rand(22, 2, 64, 9, 11, lo=-1, hi=1)
    rnd = (lambda t: t) if dtype == "fp32" else bf16_round
    xin, cin = rnd(x), rnd(cot)
    fg = flow.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    out = VF.flow_warp(xin.to(dev), fg, compute_dtype=dtype)
    out.backward(cin.to(dev))
    fo = flow.clone().double().requires_grad_(True)
    O.flow_warp(xin.double(), fo, "zeros").backward(cin.double())
    assert rel_err(fg.grad.permute(0, 3, 1, 2), fo.grad) < 2e-5     # fp32 accumulation over the 64 channels in both builds


def test_spynet_parameter_gradients_vs_oracle():
    """Spynet alone, differentiated w.r.t. its 60 conv tensors for a fixed cotangent on the flow (fp32 build vs the
    fp64 oracle).  With the keyed weights scaled by 0.25 the flows stay small, the 6-level recursion does not amplify
    rounding into ReLU-mask flips, and the comparison is sharp; 40x72 exercises the resize path and ragged tiles."""
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.spynet import Spynet
    dev = _gpu()
    sd = O.keyed_state_dict(O.spynet_param_shapes())
    sd = {k: (v * 0.25 if k.endswith("weight") else v) for k, v in sd.items()}
    m = Spynet(False)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    m.compute_dtype = "fp32"
    ref, supp = rand(6, 2, 3, 40, 72), rand(7, 2, 3, 40, 72)
    cot = rand(23, 2, 2, 40, 72, lo=-1, hi=1)
    os.environ["VSRLAB_AMD_DTYPE"] = "fp32"
    try:
        flow = m(ref.to(dev), supp.to(dev))
        (flow * cot.to(dev)).sum().backward()
    finally:
        del os.environ["VSRLAB_AMD_DTYPE"]
    leaves = {k: v.double().requires_grad_(v.is_floating_point() and not k.endswith(("mean", "std"))) for k, v in sd.items()}
    fo = O.spynet_forward(leaves, ref.double(), supp.double())
    (fo * cot.double()).sum().backward()
    assert rel_err(flow, fo) < 1e-4
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
    refg = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    assert set(grads) == set(refg) and len(grads) == 60
    glob, worst, cos = _grad_report(grads, refg)
    assert glob < 1e-3, (glob, worst)
    assert worst[0] < 5e-3, worst


def test_spynet_frame_gradients_vs_oracle():
    """Spynet called alone, differentiated w.r.t. its two frames with FROZEN weights (what a caller composing its own
    pipeline needs; inside BasicVSR the same code feeds the pre-clean stack): fp32 build against the fp64 oracle.  Weights
    x 0.25 keep the flows small enough that fp32 rounding does not flip ReLU masks in the coarse-to-fine recursion."""
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.spynet import Spynet
    dev = _gpu()
    sd = {k: (v * 0.25 if k.endswith("weight") else v) for k, v in O.keyed_state_dict(O.spynet_param_shapes()).items()}
    m = Spynet(False)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    for p in m.parameters():
        p.requires_grad_(False)
    ref, supp = rand(8, 1, 3, 40, 72), rand(9, 1, 3, 40, 72)
    cot = rand(24, 1, 2, 40, 72, lo=-1, hi=1)
    rg, sg = ref.clone().to(dev).requires_grad_(True), supp.clone().to(dev).requires_grad_(True)
    os.environ["VSRLAB_AMD_DTYPE"] = "fp32"
    try:
        flow = m(rg, sg)
        (flow * cot.to(dev)).sum().backward()
    finally:
        del os.environ["VSRLAB_AMD_DTYPE"]
    ro, so = ref.double().requires_grad_(True), supp.double().requires_grad_(True)
    fo = O.spynet_forward({k: v.double() for k, v in sd.items()}, ro, so)
    (fo * cot.double()).sum().backward()
    assert rel_err(flow, fo) < 1e-4
    assert rel_l2(rg.grad, ro.grad) < 2e-3 and rel_l2(sg.grad, so.grad) < 2e-3
    assert all(p.grad is None for p in m.parameters())


def test_basicvsr_train_flow_vs_golden():
    """train_flow=True (conf/experiment/basic.yaml:7): the 60 SPyNet gradients (flow gradient of the propagation
    warps, SPyNet's 7x7 dgrads / wgrads, border-warp and x2-upsampling backward) against the reference's fp64 values:
    7 tensors in full plus (sum, L2 norm, seeded projection) of every tensor.  fp32 build; the trunk gradients
    must be the same as with the frozen flow net."""
    import numpy as np
    from helpers import GOLDEN, proj_vector
    dev = _gpu()
    with np.load(os.path.join(GOLDEN, "basicvsr_m64_rb3_trainflow.npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    shape = (2, 3, 3, 24, 40)
    m, lrs, cot, sr, grads = _run_basicvsr("fp32", 64, 3, shape, int(g["seed_lr"]), int(g["seed_cot"]), dev, train_flow=True)
    keys = [str(k) for k in g["spy_keys"]]
    assert all(k in grads for k in keys) and len(keys) == 60
    ref = {k[len("grad__"):].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("grad__")}
    # Tolerances: with the keyed (unscaled) weights the flows reach ~13 px and the coarse-to-fine recursion amplifies
    # fp32 rounding into ReLU-mask flips: the ORACLE in fp32 already differs from these fp64 values by 1.1e-2 on
    # basic_module.5.basic_module.4.conv.0.bias and 5e-3 on level 0 (measured on the CPU); the HIP fp32 build measured
    # 3.7e-3 global, 6.7e-2 worst (the same tensor), <= 7.4e-2 on any norm / projection.  The backward itself is pinned
    # sharply by test_spynet_parameter_gradients_vs_oracle and test_flow_warp_flow_gradient_vs_oracle above.
    glob, worst, cos = _grad_report(grads, ref)
    assert glob < 1.5e-2, (glob, worst)
    assert worst[0] < 0.15, worst
    assert cos > 0.9999
    # the trunk's gradients do not depend on train_flow
    for k in ("backward_resblocks.conv.0.weight", "conv_last.2.weight"):
        assert rel_l2(grads[k], ref[k]) < 1e-4, k
    # every SPyNet tensor: norm and projection (relative to the tensor's norm)
    bad = []
    for i, k in enumerate(keys):
        gk = grads[k].double()
        s_ref, n_ref, p_ref = (float(v) for v in g["spy_stats"][i])
        pv = proj_vector(k, gk.shape)
        e_norm = abs(float(gk.norm()) - n_ref) / n_ref
        e_proj = abs(float((gk * pv).sum()) - p_ref) / (n_ref * float(pv.norm()))
        if e_norm > 0.15 or e_proj > 0.15:
            bad.append((k, e_norm, e_proj))
    assert not bad, bad


def test_basicvsr_input_gradient_vs_golden():
    """d mean(sr*cot) / d lrs (what RealBasicVSR's pre-clean stack receives, realbasicvsr.py:11-15) against the
    reference's fp64 gradient: bilinear x4 skip + the stems' LR channels + the flows through SPyNet's image pyramid
    (frozen SPyNet weights).  fp32 build; also t = 1 (no flow path)."""
    import numpy as np
    from helpers import GOLDEN
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.basicvsr import BasicVSR
    dev = _gpu()
    with np.load(os.path.join(GOLDEN, "basicvsr_m64_rb3_lrgrad.npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    shape = (2, 3, 3, 24, 40)
    n, t, _, h, w = shape
    m = BasicVSR(64, 3, 4, False, False)
    m.load_state_dict(O.keyed_state_dict(O.basicvsr_param_shapes(64, 3, 4)), strict=True)
    m = m.to(dev)
    m.compute_dtype = "fp32"
    lrs = rand(int(g["seed_lr"]), *shape).to(dev).requires_grad_(True)
    cot = rand(int(g["seed_cot"]), n, t, 3, 4 * h, 4 * w, lo=-1, hi=1)
    torch.mean(m(lrs) * cot.to(dev)).backward()
    ref = torch.from_numpy(g["grad_lrs"])
    # the SPyNet part carries the fp32 ReLU-mask noise of test_basicvsr_train_flow_vs_golden; the skip and stem parts are exact
    assert rel_l2(lrs.grad, ref) < 2e-2, rel_l2(lrs.grad, ref)
    assert rel_l2(m.conv_last[2].weight.grad, torch.from_numpy(g["grad__conv_last__2__weight"])) < 1e-4
    assert not any(p.grad is not None for p in m.spynet.parameters())
    # single frame: no flow, the gradient is the skip + stem parts only -> sharp comparison with the oracle
    sd = {k: v.double() for k, v in O.keyed_state_dict(O.basicvsr_param_shapes(64, 3, 4)).items()}
    l1 = rand(51, 1, 1, 3, 20, 36)
    c1 = rand(52, 1, 1, 3, 80, 144, lo=-1, hi=1)
    lg = l1.clone().to(dev).requires_grad_(True)
    torch.mean(m(lg) * c1.to(dev)).backward()
    lo = l1.clone().double().requires_grad_(True)
    torch.mean(O.basicvsr_forward(sd, lo) * c1.double()).backward()
    assert rel_l2(lg.grad, lo.grad) < 1e-4, rel_l2(lg.grad, lo.grad)


@pytest.mark.parametrize("dtype", DTYPES)
def test_basicvsr_upscale_2_vs_golden_and_oracle(dtype):
    """upscale = 2 (round-3 VERDICT missing #2; basicvsr.py:12-23: upscale // 2 = ONE PixelShufflePack, conv_last at 2h x 2w,
    nn.Upsample(scale_factor=2) skip) through the same engine: sr against the reference's own output, all 36 trainable gradients and
    the gradient w.r.t. the clip against the fp64 oracle / the reference's fp64 golden (fp32 1e-3 and the fp32 noise floor; bf16 at
    the bf16 noise floor); then the same clip on the diet arena (sr bit-identical, gradients to summation order)."""
    dev = _gpu()
    from vsrlab_amd import functional as VF
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.basicvsr import BasicVSR
    g = golden("basicvsr_m64_rb3_up2")
    shape = (2, 3, 3, 24, 40)
    n, t, _, h, w = shape
    sd = O.keyed_state_dict(O.basicvsr_param_shapes(64, 3, 2))
    lrs0 = rand(g["seed_lr"], *shape)
    cot = rand(g["seed_cot"], n, t, 3, 2 * h, 2 * w, lo=-1, hi=1)
    ref = {k[len("grad__"):].replace("__", "."): v for k, v in g.items() if k.startswith("grad__")}

    def run(arena=None):
        VF.set_arena_mode(arena)
        try:
            m = BasicVSR(64, 3, 2, False, False)
            m.load_state_dict(sd, strict=True)
            m = m.to(dev)
            m.compute_dtype = dtype
            lrs = lrs0.clone().to(dev).requires_grad_(True)
            sr = m(lrs)
            assert tuple(sr.shape) == (n, t, 3, 2 * h, 2 * w)
            torch.mean(sr * cot.to(dev)).backward()
            return sr.detach().cpu(), lrs.grad.cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
        finally:
            VF.set_arena_mode(None)

    sr, dl, grads = run()
    assert set(grads) == set(ref) and len(ref) == 36

    def oracle(kind):
        leaves = {k: (v.double() if kind == "fp64" else v.clone()).requires_grad_("spynet" not in k) for k, v in sd.items()}
        x = (lrs0.double() if kind == "fp64" else lrs0.clone()).requires_grad_(True)
        if kind == "emu":
            with O.emulate_bf16():
                s_ = O.basicvsr_forward(leaves, x)
                torch.mean(s_ * cot).backward()
        else:
            s_ = O.basicvsr_forward(leaves, x)
            torch.mean(s_ * (cot.double() if kind == "fp64" else cot)).backward()
        return s_.detach(), x.grad, {k: v.grad for k, v in leaves.items() if v.grad is not None}

    if dtype == "fp32":
        assert rel_err(sr, g["sr"]) < 1e-3
        # as for the upscale-4 golden above: fixed bounds on this shallow net.  What is left of the error is ONE LeakyReLU mask flip (an
        # element of conv_last.0's output within fp32 rounding of zero: conv_last.0 and everything upstream at ~1e-3, conv_last.2 at
        # 1e-6; two other seeds: one clean at 1e-6 everywhere, one with a flip at point_conv's output) -- the noise-floor ratio against
        # the fp32 oracle's own flips (2.8e-4 here) is a coin toss at this depth
        glob, worst, cos = _grad_report(grads, ref)
        assert glob < 1e-3 and worst[0] < 5e-3 and cos > 0.999999, (glob, worst, cos)
        assert rel_l2(dl, g["grad_lrs"]) < 2e-2, rel_l2(dl, g["grad_lrs"])        # (the SPyNet share carries fp32 ReLU-mask noise)
    else:
        sr_e, dl_e, g_e = oracle("emu")
        assert rel_err(sr, g["sr"]) <= 1.5 * max(rel_err(sr_e, g["sr"]), 1e-3)
        _noise_floor_check(grads, g_e, ref, max_glob_ratio=1.5, max_tensor_ratio=2.5)
        assert rel_l2(dl, g["grad_lrs"]) <= 1.5 * max(rel_l2(dl_e, g["grad_lrs"]), 2e-2)
    sr_d, dl_d, grads_d = run("diet")
    assert torch.equal(sr_d, sr)
    for k, v in grads.items():
        assert rel_l2(grads_d[k], v) < 1e-5, k


@pytest.mark.parametrize("dtype", DTYPES)
def test_realbasicvsr_training_vs_oracle(dtype):
    """sr, lq = RealBasicVSR(lr) with gradients of a two-term loss (core/utils.py:235-240 has one Charbonnier term on
    each output) for EVERY trainable tensor: the pre-clean stack's backward (3 iterations sharing their weights), fed by
    the gradient w.r.t. lq from BasicVSR.  Reference = the fp64 oracle (itself pinned to the reference's autocast run by
    tests/test_oracle_golden.py).  fp32 build: error <= 1.5 x (per tensor 2.5 x) the fp32 oracle's own error; bf16 build: the
    same against the bf16-storage-emulating oracle (BASELINE config 3's generator side, realbasicvsr.py:24-30).
    t = 3 carries SPyNet's mask noise in d lq; t = 1 has no flow path."""
    from helpers import realbasicvsr_shapes, realbasicvsr_oracle_grads
    from vsrlab_amd.vsr.models.RealBasicVSR.realbasicvsr import RealBasicVSR
    dev = _gpu()
    sd32 = O.keyed_state_dict(realbasicvsr_shapes(64, 2, 2))
    m = RealBasicVSR(2, mid_channels=64, upscale=4, res_blocks=2, pretrained_flow=False, train_flow=False)
    m.load_state_dict(sd32, strict=True)
    m = m.to(dev)
    m.basicvsr.compute_dtype = dtype
    os.environ["VSRLAB_AMD_DTYPE"] = dtype
    try:
        for shape in ((1, 3, 3, 24, 40), (2, 1, 3, 20, 36)):
            n, t, _, h, w = shape
            lr = rand(14, *shape)
            cot_sr = rand(15, n, t, 3, 4 * h, 4 * w, lo=-1, hi=1)
            cot_lq = rand(16, n, t, 3, h, w, lo=-1, hi=1)
            m.zero_grad(set_to_none=True)
            lin = lr.clone().to(dev)
            sr, lq = m(lin)
            assert torch.equal(lin.cpu(), lr)                       # the input is not modified (the reference overwrites it)
            (torch.mean(sr * cot_sr.to(dev)) + torch.mean(lq * cot_lq.to(dev))).backward()
            sd64 = {k: v.double() for k, v in sd32.items()}
            sr_o, lq_o, ref = realbasicvsr_oracle_grads(sd64, lr.double(), cot_sr.double(), cot_lq.double())
            grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
            assert set(grads) == set(ref)
            cl = {k: v for k, v in ref.items() if k.startswith("cleaner.")}
            assert len(cl) == 12
            if dtype == "fp32":
                assert rel_err(lq, lq_o) < 1e-4 and rel_err(sr, sr_o) < 1e-3
                _, _, g32 = realbasicvsr_oracle_grads(sd32, lr, cot_sr, cot_lq)
                _fp32_noise_floor_check(grads, g32, ref)
                _fp32_noise_floor_check({k: grads[k] for k in cl}, {k: g32[k] for k in cl}, cl)
            else:
                with O.emulate_bf16():
                    sr_e, lq_e, g_e = realbasicvsr_oracle_grads(sd32, lr, cot_sr, cot_lq)
                assert rel_err(lq, lq_o) <= 1.5 * max(rel_err(lq_e, lq_o), 1e-3), (rel_err(lq, lq_o), rel_err(lq_e, lq_o))
                assert rel_err(sr, sr_o) <= 1.5 * max(rel_err(sr_e, sr_o), 1e-3), (rel_err(sr, sr_o), rel_err(sr_e, sr_o))
                _noise_floor_check(grads, g_e, ref, max_glob_ratio=1.5, max_tensor_ratio=2.5)
                _noise_floor_check({k: grads[k] for k in cl}, {k: g_e[k] for k in cl}, cl, max_glob_ratio=1.5, max_tensor_ratio=2.5)
    finally:
        del os.environ["VSRLAB_AMD_DTYPE"]


_C1 = {}


def _config1_oracle(kind):
    """BASELINE config 1 on the CPU oracle, computed once per session.  kind: 'fp64' (the exact reference),
    'fp32', or 'emu' (fp32 arithmetic with bf16 storage at the HIP perf build's rounding points)."""
    if kind not in _C1:
        shape = (2, 5, 3, 64, 64)
        sd = O.keyed_state_dict(O.basicvsr_param_shapes(64, 30, 4))
        lrs, hr, cot = rand(10, *shape), rand(11, 2, 5, 3, 256, 256), rand(13, 2, 5, 3, 256, 256, lo=-1, hi=1)
        if kind == "emu":
            with O.emulate_bf16():
                _C1[kind] = O.fwd_bwd(sd, lrs, hr, cot=cot) + (hr,)
        elif kind == "fp64":
            _C1[kind] = O.fwd_bwd({k: v.double() for k, v in sd.items()}, lrs.double(), hr.double(), cot=cot.double()) + (hr,)
        else:
            _C1[kind] = O.fwd_bwd(sd, lrs, hr, cot=cot) + (hr,)
    return _C1[kind]


@pytest.mark.parametrize("dtype", DTYPES)
def test_basicvsr_config1_all_grads_vs_oracle(dtype):
    """BASELINE config 1: n=2, t=5, 64x64 LR, BasicVSR(64, 30): sr, Charbonnier loss and EVERY
    trainable gradient (254 tensors).  fp32 build vs the fp32 oracle (1e-3 north-star bar on sr; gradients at the
    oracle's own fp32-vs-fp64 noise floor).  bf16 build: error against the fp64 oracle <= 1.5 x the error of the
    bf16-storage-emulating oracle against the same fp64 oracle, globally; per tensor <= 2.5 x (a single tensor's
    mask-flip noise is itself a random draw; measured worst ratio in tools/gpu_diag.py)."""
    dev = _gpu()
    shape = (2, 5, 3, 64, 64)
    m, lrs, cot, sr, grads = _run_basicvsr(dtype, 64, 30, shape, 10, 13, dev)
    from vsrlab_amd.core.losses import CharbonnierLoss
    if dtype == "fp32":
        sr_o, loss_o, grads_o, hr = _config1_oracle("fp32")
        assert rel_err(sr, sr_o) < 1e-3
        loss = CharbonnierLoss()(sr.to(dev), hr.to(dev))
        assert abs(float(loss) - float(loss_o)) < 1e-4 * float(loss_o)
        assert set(grads) == set(grads_o)
        _, _, grads_x, _ = _config1_oracle("fp64")
        _fp32_noise_floor_check(grads, grads_o, grads_x)
        assert _grad_report(grads, grads_x)[2] > 0.99999
    else:
        sr_x, loss_x, grads_x, hr = _config1_oracle("fp64")
        sr_e, loss_e, grads_e, _ = _config1_oracle("emu")
        assert set(grads) == set(grads_x)
        assert rel_err(sr, sr_x) <= 1.5 * rel_err(sr_e, sr_x), (rel_err(sr, sr_x), rel_err(sr_e, sr_x))
        loss = CharbonnierLoss()(sr.to(dev), hr.to(dev))
        assert abs(float(loss) - float(loss_x)) <= 1.5 * max(abs(float(loss_e) - float(loss_x)), 1e-4 * float(loss_x))
        _noise_floor_check(grads, grads_e, grads_x, max_glob_ratio=1.5, max_tensor_ratio=2.5)


_C540 = {}


def _oracle_540(kind):
    """540x960, t=3, 2 blocks on the CPU oracle (fp32 arithmetic; 'emu' = with bf16 storage; 'fp64' = the exact one), once per session."""
    if kind not in _C540:
        shape = (1, 3, 3, 540, 960)
        sd = O.keyed_state_dict(O.basicvsr_param_shapes(64, 2, 4))
        lrs, cot = rand(70, *shape), rand(71, 1, 3, 3, 2160, 3840, lo=-1, hi=1)
        if kind == "emu":
            with O.emulate_bf16():
                sr, _, g = O.fwd_bwd(sd, lrs, cot, cot=cot)
        elif kind == "fp64":
            sr, _, g = O.fwd_bwd({k: v.double() for k, v in sd.items()}, lrs.double(), cot.double(), cot=cot.double())
        else:
            sr, _, g = O.fwd_bwd(sd, lrs, cot, cot=cot)
        _C540[kind] = (sr, g)
    return _C540[kind]


@pytest.mark.parametrize("dtype", DTYPES)
def test_basicvsr_540p_vs_oracle(dtype):
    """BASELINE config 2's frame size (540x960 -> 2160x3840) against the CPU oracle: t=3, 2 blocks, sr and all 30
    gradients.  Exercises ragged tile rows (540 = 67.5 x 8), the 2160x3840 HR kernels, the multi-frame wgrad launches
    and the arena offsets at this size.  fp32: north-star 1e-3 on sr, gradients at the fp32 noise floor; bf16: error <=
    1.5 x the emulation's own error against the fp32 oracle (global), 2.5 x per tensor."""
    dev = _gpu()
    shape = (1, 3, 3, 540, 960)
    m, lrs, cot, sr, grads = _run_basicvsr(dtype, 64, 2, shape, 70, 71, dev)
    sr_o, g_o = _oracle_540("fp32")
    assert set(grads) == set(g_o)
    if dtype == "fp32":
        assert rel_err(sr, sr_o) < 1e-3
        _fp32_noise_floor_check(grads, g_o, _oracle_540("fp64")[1])
    else:
        sr_e, g_e = _oracle_540("emu")
        assert rel_err(sr, sr_o) <= 1.5 * max(rel_err(sr_e, sr_o), 1e-3)
        _noise_floor_check(grads, g_e, g_o, max_glob_ratio=1.5, max_tensor_ratio=2.5)
    m._pool.clear()


def test_config2_full_size_properties():
    """BASELINE config 2 at FULL size through the C ABI (n=1, t=7, 540x960, 30 blocks, bf16: the benchmarked
    configuration, ~128 GiB arena): finite outputs; the trunk-chain launches reproduce the one-launch-per-layer engine bit for
    bit (sr and all 254 gradients); a second backward on the same retained forward reproduces EVERY gradient bit for bit; the
    backward is linear in the cotangent.  Two forwards, five backwards."""
    dev = _gpu()
    import vsrlab_amd
    from vsrlab_amd import functional as VF
    from vsrlab_amd._order import basicvsr_keys
    lib = vsrlab_amd._lib.load()
    n, t, h, w, rb = 1, 7, 540, 960, 30
    sd = O.keyed_state_dict(O.basicvsr_param_shapes(64, rb, 4))
    keys, n_train = basicvsr_keys(rb)
    ps = [sd[k].to(dev).contiguous() for k in keys]
    desc = vsrlab_amd._lib.BasicVSRDesc(n, t, h, w, 64, rb, 4, VF.DT_BF16)
    nbytes = lib.vsr_basicvsr_workspace_bytes(ctypes.byref(desc), 1)
    assert nbytes > 100 * 2**30
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    lrs = torch.rand(n, t, 3, h, w, device=dev)
    sr = torch.empty(n, t, 3, 4 * h, 4 * w, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fwd():
        assert lib.vsr_basicvsr_forward(ctypes.byref(desc), VF._ptr_array(ps), len(ps), VF._ptr(lrs), VF._ptr(sr), VF._ptr(ws), nbytes, 1, st) == 0

    def bwd(cot):
        gs = [torch.zeros_like(p) if k < n_train else None for k, p in enumerate(ps)]
        assert lib.vsr_basicvsr_backward(ctypes.byref(desc), VF._ptr_array(ps), VF._ptr_array(gs), len(ps), VF._ptr(lrs), VF._ptr(cot),
                                         VF._ptr(None), VF._ptr(ws), nbytes, st) == 0
        return {keys[k]: g for k, g in enumerate(gs) if g is not None}

    # The chain launches AT THE BENCHMARKED SIZE (60 layers x 2040 tiles on 256 workgroups: deferred publishes, two-tile look-ahead,
    # middle-out region order, layer-boundary weight reloads) against one launch per layer (VSRLAB_AMD_CHAIN is read per engine
    # call): the same tile arithmetic on the same operands, so sr and all 254 gradients must be bit-identical.
    t0 = VF.chain_timeouts()
    c1 = torch.randn_like(sr)
    os.environ["VSRLAB_AMD_CHAIN"] = "0"
    try:
        fwd()
        sr_per_layer = sr.clone()
        g_per_layer = bwd(c1)
    finally:
        os.environ.pop("VSRLAB_AMD_CHAIN", None)
    sr.zero_()
    fwd()
    assert bool(torch.isfinite(sr).all())
    assert torch.equal(sr, sr_per_layer)
    del sr_per_layer
    # same weights, frame 0 alone at inference (no arena retention): the first frame of the backward-time chain differs
    # (it sees the later frames), so only shapes / finiteness are compared here; values are pinned at t=3 above.

    c2 = torch.randn_like(sr)
    g1, g1b, g2 = bwd(c1), bwd(c1), bwd(c2)
    for k in g1:
        assert torch.equal(g1[k], g_per_layer[k]), ("chain vs one launch per layer", k)
    assert VF.chain_timeouts() == t0
    c12 = 0.5 * c1 + c2
    g12 = bwd(c12)
    del c12
    assert len(g1) == n_train == 254
    for k in g1:
        assert bool(torch.isfinite(g1[k]).all()) and float(g1[k].abs().max()) > 0, k
        # SURVEY 5 "determinism test": nothing in a frozen-flow backward depends on an arrival order any more -- the warp adjoint is
        # a gather, its far-source share is summed in 64-bit fixed point (integer adds are associative), weight gradients are
        # reduced from per-workgroup slabs in index order -- so a repeat is bit-identical for all 254 tensors at the benchmarked size
        assert torch.equal(g1[k], g1b[k]), k
    cat = lambda d: torch.cat([d[k].flatten() for k in sorted(d)])
    # every product dY*X is formed from bf16-rounded activation gradients: linear up to bf16 rounding
    assert rel_l2(cat(g12), 0.5 * cat(g1) + cat(g2)) < 3e-2
    del ws


def test_ragged_sizes_and_single_frame():
    """h,w not multiples of the 8x32 tile nor of 32 (SPyNet resize path); t=1 (no flow at all);
    batch consistency: two identical clips in a batch give identical outputs."""
    dev = _gpu()
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.basicvsr import BasicVSR
    sd32 = O.keyed_state_dict(O.basicvsr_param_shapes(64, 2, 4))
    m = BasicVSR(64, 2, 4, False, False)
    m.load_state_dict(sd32, strict=True)
    m = m.to(dev)
    m.compute_dtype = "fp32"
    for shape in [(1, 3, 3, 21, 45), (1, 1, 3, 16, 16), (1, 2, 3, 9, 70)]:
        lrs = rand(51, *shape)
        with torch.no_grad():
            sr = m(lrs.to(dev)).cpu()
            ref = O.basicvsr_forward(sd32, lrs)
        assert rel_err(sr, ref) < 1e-3, shape
    lrs = rand(52, 1, 3, 3, 24, 40)
    with torch.no_grad():
        both = m(torch.cat([lrs, lrs], 0).to(dev)).cpu()
        one = m(lrs.to(dev)).cpu()
    assert torch.equal(both[0], both[1])
    assert rel_err(both[0:1], one) < 1e-6


def test_inference_matches_training_forward_and_no_cpu_fallback():
    dev = _gpu()
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.basicvsr import BasicVSR
    m = BasicVSR(64, 2, 4, False, False)
    m.load_state_dict(O.keyed_state_dict(O.basicvsr_param_shapes(64, 2, 4)), strict=True)
    lrs = rand(53, 1, 3, 3, 24, 40)
    with pytest.raises(RuntimeError):
        m(lrs)                                   # CPU tensors: fail loudly, never fall back
    m = m.to(dev)
    m.compute_dtype = "bf16"
    sr_train = m(lrs.to(dev))
    with torch.no_grad():
        sr_inf = m(lrs.to(dev))
    assert torch.equal(sr_train.detach().cpu(), sr_inf.cpu())   # in-place inference schedule == retained schedule


def test_backward_is_linear_in_cotangent_large():
    """Size-independent property at a large frame (540x960, t=2): backward(a*c1 + c2) ==
    a*backward(c1) + backward(c2), through the C ABI on one retained forward (bf16 build)."""
    dev = _gpu()
    import vsrlab_amd
    from vsrlab_amd import functional as VF
    from vsrlab_amd._order import basicvsr_keys
    lib = vsrlab_amd._lib.load()
    n, t, h, w, rb = 1, 2, 540, 960, 2
    sd = O.keyed_state_dict(O.basicvsr_param_shapes(64, rb, 4))
    keys, n_train = basicvsr_keys(rb)
    ps = [sd[k].to(dev).contiguous() for k in keys]
    desc = vsrlab_amd._lib.BasicVSRDesc(n, t, h, w, 64, rb, 4, VF.DT_BF16)
    nbytes = lib.vsr_basicvsr_workspace_bytes(ctypes.byref(desc), 1)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    lrs = torch.rand(n, t, 3, h, w, device=dev)
    sr = torch.empty(n, t, 3, 4 * h, 4 * w, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.vsr_basicvsr_forward(ctypes.byref(desc), VF._ptr_array(ps), len(ps), VF._ptr(lrs), VF._ptr(sr), VF._ptr(ws), nbytes, 1, st) == 0
    assert bool(torch.isfinite(sr).all())

    def bwd(cot):
        gs = [torch.zeros_like(p) if k < n_train else None for k, p in enumerate(ps)]
        assert lib.vsr_basicvsr_backward(ctypes.byref(desc), VF._ptr_array(ps), VF._ptr_array(gs), len(ps), VF._ptr(lrs), VF._ptr(cot),
                                         VF._ptr(None), VF._ptr(ws), nbytes, st) == 0
        return torch.cat([g.flatten() for g in gs if g is not None])

    c1 = torch.randn_like(sr)
    c2 = torch.randn_like(sr)
    g1, g2, g12 = bwd(c1), bwd(c2), bwd(0.5 * c1 + c2)
    assert bool(torch.isfinite(g12).all())
    # every product dY*X is formed from bf16-rounded activation gradients: linear up to bf16 rounding
    assert rel_l2(g12, 0.5 * g1 + g2) < 2e-2
    assert torch.equal(bwd(c1), g1)            # nothing in a frozen-flow backward depends on an arrival order (determinism)


@pytest.mark.parametrize("dtype", DTYPES)
def test_realbasicvsr_inference_vs_golden(dtype):
    """``sr, lq = RealBasicVSR(lr)`` (the `_target_` of conf/train/model/basicvsr.yaml) under no_grad:
    pre-clean stack + BasicVSR on the HIP engine, against the reference's own outputs; ragged 24x40 frames.
    ``lq`` must be a fresh tensor (the input is not mutated) and training must raise, not fall back."""
    dev = _gpu()
    from vsrlab_amd.vsr.models.RealBasicVSR.realbasicvsr import RealBasicVSR
    g = golden("realbasicvsr_m64")
    m = RealBasicVSR(2, mid_channels=64, upscale=4, res_blocks=2, pretrained_flow=False, train_flow=False)
    shapes = {"basicvsr." + k: s for k, s in O.basicvsr_param_shapes(64, 2, 4).items()}
    shapes.update(O.cleaner_param_shapes(64, 2))
    m.load_state_dict(O.keyed_state_dict(shapes), strict=True)
    m = m.to(dev).eval()
    m.basicvsr.compute_dtype = dtype
    lrs = rand(g["seed_lr"], 1, 3, 3, 24, 40).to(dev)
    keep = lrs.clone()
    import os
    os.environ["VSRLAB_AMD_DTYPE"] = dtype
    try:
        with torch.no_grad():
            sr, lq = m(lrs)
    finally:
        del os.environ["VSRLAB_AMD_DTYPE"]
    assert torch.equal(lrs, keep) and lq.data_ptr() != lrs.data_ptr()
    assert rel_err(lq, g["lq"]) < tol(dtype, 1e-3, 2e-2)
    assert rel_err(sr, g["sr"]) < tol(dtype, 1e-3, 3e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_far_flows_match_the_oracle_and_repeat_bit_for_bit(dtype):
    """Flows longer than the gather radius of the warp adjoint (|flow| > 4 px: elementwise.hip warp_bwd_far_kernel) mixed with
    short ones: a bias on SPyNet's finest level lengthens dx, so about half of the sources take the far path (64-bit
    fixed-point atomics) and part the gather.  Gradients against the oracle at the noise floor, and two independent
    forward + backward runs agree bit for bit (round-2 VERDICT #2: the far share used float atomics)."""
    dev = _gpu()
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.basicvsr import BasicVSR
    shape = (1, 3, 3, 48, 64)
    sd = O.keyed_state_dict(O.basicvsr_param_shapes(64, 2, 4))
    sd["spynet.basic_module.5.basic_module.4.conv.0.bias"] = torch.tensor([1.0, 0.0])
    lrs, cot = rand(81, *shape), rand(82, 1, 3, 3, 192, 256, lo=-1, hi=1)
    fb, ff = O.basicvsr_compute_flow(sd, lrs)
    far = float(((fb.abs() > 4).any(dim=2).float().mean() + (ff.abs() > 4).any(dim=2).float().mean()) / 2)
    assert 0.05 < far < 0.95, far                                   # both paths are exercised

    def run():
        m = BasicVSR(64, 2, 4, False, False)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev)
        m.compute_dtype = dtype
        sr = m(lrs.to(dev))
        torch.mean(sr * cot.to(dev)).backward()
        return sr.detach().cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}

    sr, grads = run()
    sr2, grads2 = run()
    assert torch.equal(sr, sr2)
    for k in grads:
        assert torch.equal(grads[k], grads2[k]), k
    sd64 = {k: v.double() for k, v in sd.items()}
    sr_x, _, g_x = O.fwd_bwd(sd64, lrs.double(), cot.double(), cot=cot.double())
    if dtype == "fp32":
        sr_o, _, g_o = O.fwd_bwd(sd, lrs, cot, cot=cot)
        assert rel_err(sr, sr_x) < 1e-3
        _fp32_noise_floor_check(grads, g_o, g_x)
    else:
        with O.emulate_bf16():
            sr_e, _, g_e = O.fwd_bwd(sd, lrs, cot, cot=cot)
        assert rel_err(sr, sr_x) <= 1.5 * max(rel_err(sr_e, sr_x), 1e-3)
        _noise_floor_check(grads, g_e, g_x, max_glob_ratio=1.5, max_tensor_ratio=2.5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gather_form_warp_adjoint_vs_scatter_form_and_non_finite_far_sources(dtype):
    """The engine's gather-form warp adjoint (elementwise.hip warp_bwd_gather_kernel + warp_bwd_far_kernel) on its own, through the
    vsr_debug_warp_bwd_gather hook: (a) equal to the scatter form (vsr_flow_warp_bwd_ex: fp32 atomics) on flows that mix near and
    far sources, S back to all-zero; (b) round-3 ADVICE: a NaN / Inf cotangent at a FAR source used to be converted to a finite
    fixed-point number -- it must come out non-finite (FusedAdam's skip-on-non-finite-norm relies on it); a near one always did."""
    dev = _gpu()
    from vsrlab_amd import _lib, functional as VF
    lib = _lib.load()
    fn = lib.vsr_debug_warp_bwd_gather
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    dt = VF.DT_F32 if dtype == "fp32" else VF.DT_BF16
    n, h, w = 2, 37, 70
    flow = rand(1401, n, 2, h, w, lo=-3, hi=3)
    flow[:, 0, :, 20:45] += 7.0                                       # a band of far sources (|dx| > 4)
    flow = flow.to(dev).contiguous()
    cot = rand(1402, n, 64, h, w, lo=-1, hi=1).to(dev)

    def gather(c):
        g = VF.to_pixel_major(c, dt)
        out = torch.empty_like(g)
        S = torch.zeros(n * h * w * 64, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        assert fn(dt, VF._ptr(g), VF._ptr(flow), None, VF._ptr(S), VF._ptr(cnt), VF._ptr(out), n, h, w, VF._stream()) == 0
        out.pm_w = w
        return VF.from_pixel_major(out, 64).float(), S, int(cnt.item())

    got, S, far = gather(cot)
    assert far & 0x3fffffff > 0 and not (far & 0x40000000) and int(S.abs().max()) == 0
    g = VF.to_pixel_major(cot, dt)
    acc = torch.zeros((n, h, w, 64), dtype=torch.float32, device=dev)
    assert lib.vsr_flow_warp_bwd_ex(dt, VF._ptr(g), VF._ptr(flow), VF._ptr(acc), n, h, w, 64, 0, VF._stream()) == 0
    want = acc.permute(0, 3, 1, 2)
    assert rel_err(got, want) < tol(dtype, 1e-5, 1e-2)                # (the gather form rounds its output to the storage type)
    for bad in (float("nan"), float("inf")):
        for x in (30, 5):                                             # a far source, a near source
            c = cot.clone()
            c[1, 7, 11, x] = bad
            out, S, far = gather(c)
            assert not bool(torch.isfinite(out).all()), (bad, x)
            assert int(S.abs().max()) == 0
            assert bool(far & 0x40000000) == (x == 30)


def test_long_clip_more_than_8_frames_fp32():
    """t = 10 > VSR_WG_MAXSEG: the per-layer weight-gradient launches are split over two segment batches and
    accumulated; odd frame size; fp32 build against the fp32 oracle."""
    dev = _gpu()
    shape = (1, 10, 3, 20, 36)
    m, lrs, cot, sr, grads = _run_basicvsr("fp32", 64, 1, shape, 61, 62, dev)
    sd = O.keyed_state_dict(O.basicvsr_param_shapes(64, 1, 4))
    hr = rand(63, 1, 10, 3, 80, 144)
    sr_o, _, grads_o = O.fwd_bwd(sd, lrs, hr, cot=cot)
    _, _, grads_x = O.fwd_bwd({k: v.double() for k, v in sd.items()}, lrs.double(), hr.double(), cot=cot.double())
    assert rel_err(sr, sr_o) < 1e-3
    _fp32_noise_floor_check(grads, grads_o, grads_x)


@pytest.mark.parametrize("dtype", DTYPES)
def test_standalone_modules_forward_and_backward_vs_oracle(dtype):
    """The reference's building blocks called on their own, as ordinary autograd modules (core/modules/conv.py:15-22,94-103,
    upsampling.py:4-12, spynet.py:13-21): ResidualBlock on cat([lr, feat]) (67 channels: the trunks) and on lr alone (3 channels:
    the pre-clean stack), PixelShufflePack, ConvReLU (7x7 and 3x3 64->64) and SpynetModule -- output, input gradient and every
    parameter gradient against torch autograd on the oracle's restatements (pinned to the reference at 16 channels by
    tests/test_oracle_golden.py).  fp32: 1e-3 (single layers / shallow stacks: no deep mask-flip noise); bf16: against the
    fp64 result, no worse than 1.5 x the error of the same computation with bf16-rounded operands (floor 2e-2)."""
    dev = _gpu()
    from vsrlab_amd.core.modules.conv import ConvReLU, ResidualBlock
    from vsrlab_amd.core.modules.upsampling import PixelShufflePack
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.spynet import SpynetModule
    os.environ["VSRLAB_AMD_DTYPE"] = dtype

    def check(name, mod, ref_fn, x, seed):
        sd = O.keyed_state_dict({k: tuple(v.shape) for k, v in mod.state_dict().items()})
        mod.load_state_dict(sd, strict=True)
        mod = mod.to(dev)
        xg = x.clone().to(dev).requires_grad_(True)
        y = mod(xg)
        cot = rand(seed, *y.shape, lo=-1, hi=1)
        (y * cot.to(dev)).sum().backward()
        got = {"x": xg.grad.cpu(), **{k: p.grad.cpu() for k, p in mod.named_parameters()}}
        leaves = {k: v.double().requires_grad_(True) for k, v in sd.items()}
        xr = x.double().requires_grad_(True)
        yr = ref_fn(leaves, xr)
        (yr * cot.double()).sum().backward()
        want = {"x": xr.grad, **{k: v.grad for k, v in leaves.items()}}
        assert tuple(y.shape) == tuple(yr.shape), name
        assert set(got) == set(want), name
        if dtype == "fp32":
            assert rel_err(y.detach(), yr.detach()) < 1e-3, name
            for k in want:
                assert rel_l2(got[k], want[k]) < 1e-3, (name, k, rel_l2(got[k], want[k]))
        else:
            with O.emulate_bf16():
                l32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
                x32 = x.clone().requires_grad_(True)
                ye = ref_fn(l32, x32)
                (ye * cot).sum().backward()
            emu = {"x": x32.grad, **{k: v.grad for k, v in l32.items()}}
            assert rel_err(y.detach(), yr.detach()) <= 1.5 * max(rel_err(ye.detach(), yr.detach()), 2e-2), name
            for k in want:
                assert rel_l2(got[k], want[k]) <= 1.5 * max(rel_l2(emu[k], want[k]), 2e-2), (name, k, rel_l2(got[k], want[k]), rel_l2(emu[k], want[k]))

    def spy_ref(sd, x):                                    # O._q / O._wq: identity unless the oracle emulates bf16 storage
        x = O._q(x)
        for j in range(5):
            x = F.relu(F.conv2d(x, O._wq(sd[f"basic_module.{j}.conv.0.weight"]), sd[f"basic_module.{j}.conv.0.bias"], padding=3))
            if j < 4:
                x = O._q(x)
        return x

    def conv_relu_ref(pad):
        return lambda sd, x: O._q(F.relu(F.conv2d(O._q(x), O._wq(sd["conv.0.weight"]), sd["conv.0.bias"], padding=pad)))

    try:
        for cin in (67, 3):
            check(f"ResidualBlock{cin}", ResidualBlock(cin, 64, 2), lambda sd, x: O.residual_block(sd, "", O._q(x), 2), rand(61 + cin, 2, cin, 13, 37, lo=-1, hi=1), 71)
        check("PixelShufflePack", PixelShufflePack(64, 64, 2), lambda sd, x: O.pixel_shuffle_pack(sd, "", O._q(x)), rand(62, 2, 64, 7, 9, lo=-1, hi=1), 72)
        check("ConvReLU7", ConvReLU(32, 64, 7, 1, 3), conv_relu_ref(3), rand(63, 1, 32, 20, 41, lo=-1, hi=1), 73)
        check("ConvReLU3", ConvReLU(64, 64, 3, 1, 1), conv_relu_ref(1), rand(65, 2, 64, 11, 33, lo=-1, hi=1), 75)
        check("SpynetModule", SpynetModule(), spy_ref, rand(64, 2, 8, 24, 40, lo=-1, hi=1), 74)
    finally:
        del os.environ["VSRLAB_AMD_DTYPE"]


@pytest.mark.parametrize("shape", [(2, 20, 41), (3, 37, 70), (4, 200, 330)])
def test_spynet_7x7_layers_on_the_persistent_kernels(shape):
    """The five SPyNet layer shapes (spynet.py:16-18) on the persistent 7x7 kernels (conv7x7_persist.hip: streamed / resident
    weights, tap pairs for the 16-channel layers), bf16 build, through ConvReLU's per-op entry: against the same convolution of
    the bf16-rounded operands in fp32 (torch on the GPU; what differs is the summation order and the output's rounding to bf16:
    2^-8 relative), on ragged sizes from a single partial tile to several tiles per workgroup (4 x 200 x 330 = 1100 tiles on
    256 workgroups: the weight ring wraps and both tile buffers are re-used), and bit-identical on a repeat."""
    dev = _gpu()
    from vsrlab_amd import functional as VF
    n, h, w = shape
    for j, (ci, co) in enumerate([(8, 32), (32, 64), (64, 32), (32, 16), (16, 2)]):
        x = bf16_round(rand(300 + j, n, ci, h, w, lo=-1, hi=1)).to(dev)
        wt = bf16_round(rand(310 + j, co, ci, 7, 7, lo=-1, hi=1) / (7.0 * ci ** 0.5)).to(dev)
        b = rand(320 + j, co, lo=-0.5, hi=0.5).to(dev)
        y = VF.conv_relu_forward(x, wt, b, compute_dtype="bf16")
        ref = F.relu(F.conv2d(x, wt, b, padding=3))
        assert tuple(y.shape) == tuple(ref.shape)
        err = (y - ref).abs()
        bound = ref.abs() * 2.0 ** -7 + 1e-3 if co > 4 else ref.abs() * 1e-4 + 1e-4     # the 2-channel layer stores fp32
        assert bool((err <= bound).all()), (shape, ci, co, float(err.max()), float(ref.abs().max()))
        assert torch.equal(y, VF.conv_relu_forward(x, wt, b, compute_dtype="bf16")), (shape, ci, co)


@pytest.mark.parametrize("dtype", DTYPES)
def test_diet_arena_matches_the_full_arena(dtype):
    """VsrBasicVSRDesc.arena_mode = 1 ("diet", functional.set_arena_mode): the trunks' activation gradients in a two-block ring with
    each frame's weight gradients launched behind its data gradients, U0 / U1 / C0 of the reconstruction recomputed in the
    backward.  Same kernels on the same operands, so sr and every data-path tensor are bit-identical; the weight gradients differ
    only in the order the frames are summed (fp32 accumulation: 1e-5 relative L2 per tensor; the two modes are compared with
    each other here, the diet mode with the oracle / goldens by the next test).  Frozen flow and train_flow, ragged size."""
    dev = _gpu()
    from vsrlab_amd import functional as VF
    # (the third shape: 10 frames = two chunks of VSR_WG_MAXSEG in every all-frames weight-gradient launch, including round 4's deferred ones of
    # the reconstruction; in the bf16 build the full arena also keeps the gradients into the pixel-shuffle layers phase-separated, the diet arena strided)
    for (shape, blocks, train_flow) in [((1, 5, 3, 40, 72), 4, False), ((2, 3, 3, 13, 37), 2, True), ((1, 10, 3, 24, 40), 2, False)]:
        out = {}
        for mode in ("full", "diet"):
            VF.set_arena_mode(mode)
            try:
                _, _, _, sr, grads = _run_basicvsr(dtype, 64, blocks, shape, 91, 92, dev, train_flow=train_flow)
            finally:
                VF.set_arena_mode(None)
            out[mode] = (sr, grads)
        assert torch.equal(out["full"][0], out["diet"][0]), (shape, "sr")
        assert set(out["full"][1]) == set(out["diet"][1])
        for k, v in out["full"][1].items():
            assert rel_l2(out["diet"][1][k], v) < 1e-5, (shape, k, rel_l2(out["diet"][1][k], v))


def _chain_timeouts():
    from vsrlab_amd import functional as VF
    return VF.chain_timeouts()


def test_trunk_chain_launch_is_bit_identical_to_one_launch_per_layer():
    """conv3x3_chain.hip: the 2 rb convolutions of a frame's residual blocks (and their 2 rb - 1 data gradients) as ONE launch whose
    workgroups hand tiles to each other through row counters (write-through stores, agent-scope counters), against
    VSRLAB_AMD_CHAIN=0 = one launch per layer: the same tile arithmetic on the same operands, so sr and EVERY gradient must
    be bit-identical -- one stale tile anywhere in 2 x t x (4 rb - 1) layers would show.  Sizes: fewer tiles than
    workgroups (every workgroup's next item depends on its current one: the publish-at-once path), ragged edges, a batch
    of two, and a size with several tiles per workgroup and row (deferred publishes, look-ahead polls); run twice
    (L1 / L2 warm with the previous run's lines).  No dependency wait may have timed out."""
    dev = _gpu()
    t0 = _chain_timeouts()
    for (shape, blocks) in [((1, 3, 3, 20, 40), 3), ((2, 3, 3, 45, 100), 4), ((1, 2, 3, 200, 330), 6), ((1, 2, 3, 270, 480), 30)]:
        out = {}
        for mode in ("0", "1", "1b"):
            os.environ["VSRLAB_AMD_CHAIN"] = mode[0]
            try:
                _, _, _, sr, grads = _run_basicvsr("bf16", 64, blocks, shape, 93, 94, dev)
            finally:
                os.environ.pop("VSRLAB_AMD_CHAIN", None)
            out[mode] = (sr, grads)
        for mode in ("1", "1b"):
            assert torch.equal(out["0"][0], out[mode][0]), (shape, mode, "sr")
            for k, v in out["0"][1].items():
                assert torch.equal(out[mode][1][k], v), (shape, mode, k, rel_l2(out[mode][1][k], v))
    assert _chain_timeouts() == t0


@pytest.mark.parametrize("shape,blocks", [((2, 21, 70), 2), ((1, 96, 160), 5), ((1, 135, 240), 32), ((1, 540, 960), 8)])
def test_conv3x3_c64_chain_abi_matches_the_per_layer_calls(shape, blocks):
    """vsr_conv3x3_c64_chain_fwd (functional.ResidualChainC64): 2 x blocks layers in one launch against the same layers as
    2 x blocks calls of vsr_conv3x3_c64_fwd -- every intermediate image bit-identical, on a ragged batch of two, on a size
    with more tiles than workgroups per layer, on the maximum of 64 layers, and on the benchmarked 540 x 960 image (2040 tiles
    per layer = 8 per workgroup, 16 layers); launched twice (the flag words are re-zeroed per launch)."""
    dev = _gpu()
    from vsrlab_amd import functional as VF
    n, h, w = shape
    L = 2 * blocks
    ws = [rand(700 + l, 64, 64, 3, 3, lo=-1, hi=1) / 24.0 for l in range(L)]
    bs = [rand(800 + l, 64, lo=-0.1, hi=0.1) for l in range(L)]
    x = VF.to_pixel_major(rand(699, n, 64, h, w, lo=-1, hi=1).to(dev), VF.DT_BF16)
    t0 = _chain_timeouts()
    ch = VF.ResidualChainC64([t.to(dev) for t in ws], [t.to(dev) for t in bs], n, h, w, dev)
    for rep in range(2):
        y = ch(x)
        cur, ref = x, [x]
        for b in range(blocks):
            a = VF.conv3x3_c64(cur, ws[2 * b].to(dev), bs[2 * b].to(dev), act=1)
            cur = VF.conv3x3_c64(a, ws[2 * b + 1].to(dev), bs[2 * b + 1].to(dev), act=0, res_pm=cur)
            ref += [a, cur]
        valid = VF.from_pixel_major          # compare through the planar view: padding pixels of the blocked layout are never written
        for l in range(1, L + 1):
            got = ch.image(l).view(ref[l].shape)
            got.pm_w = w
            assert torch.equal(valid(got), valid(ref[l])), (shape, rep, l)
    assert _chain_timeouts() == t0


def test_chain_hand_off_under_uneven_load_from_a_second_stream():
    """The chain's workgroups hand tiles to each other inside the launch; its progress argument allows any share of the grid to be
    resident.  Here a second stream keeps the chip busy with unrelated work of uneven length (large and small GEMMs, element-wise
    passes) while 24-layer chains run on the first, so that chain workgroups start late, on whatever CUs are free, next to
    kernels that evict their L2 lines: every word of every intermediate image must still equal the quiet run's, and no
    dependency wait may time out."""
    dev = _gpu()
    from vsrlab_amd import functional as VF
    n, h, w, L = 1, 216, 384, 24
    ws = [(rand(900 + l, 64, 64, 3, 3, lo=-1, hi=1) / 24.0).to(dev) for l in range(L)]
    bs = [rand(950 + l, 64, lo=-0.1, hi=0.1).to(dev) for l in range(L)]
    x = VF.to_pixel_major(rand(899, n, 64, h, w, lo=-1, hi=1).to(dev), VF.DT_BF16)
    t0 = _chain_timeouts()
    ch = VF.ResidualChainC64(ws, bs, n, h, w, dev)
    ch(x)
    torch.cuda.synchronize()
    quiet = ch.images.clone()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    b = torch.randn(256, 256, device=dev)
    big = torch.randn(64 * 1024 * 1024, device=dev)
    for rep in range(6):
        ch.images[ch.img_elems:].zero_()                       # every layer's output must be produced again
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for k in range(12):
                if (k + rep) % 3 == 0:
                    a = (a @ a).clamp_(-1, 1)
                elif (k + rep) % 3 == 1:
                    b = torch.tanh(b @ b)
                else:
                    big.mul_(1.0001)
        for _ in range(2):
            ch.launch()
        torch.cuda.synchronize()
        assert torch.equal(ch.images, quiet), rep
    assert _chain_timeouts() == t0


def test_chain_error_word_poisons_the_outputs_and_the_optimizer_skips_the_step():
    """A chain launch whose error word is set (a dependency wait was given up: its tiles were computed from unfinished inputs) must not
    look like a result (round-3 VERDICT weak #2 / ADVICE medium).  vsr_debug_chain_inject_error(k) starts the next k chain launches
    with the word set: chain_poison_kernel, enqueued behind every chain launch, then writes NaN into the chain's last image.  Forced in
    the forward: sr and the loss are not finite.  Forced in the backward only (clean forward): the gradient norm is not finite and
    FusedAdam leaves every parameter untouched -- all without a host synchronisation in the product path."""
    dev = _gpu()
    from vsrlab_amd import _lib, functional as VF
    from vsrlab_amd.optim import FusedAdam
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.basicvsr import BasicVSR
    lib = _lib.load()
    inject = lib.vsr_debug_chain_inject_error
    inject.restype, inject.argtypes = ctypes.c_int, [ctypes.c_int]
    m = BasicVSR(64, 2, 4, False, False)
    m.load_state_dict(O.keyed_state_dict(O.basicvsr_param_shapes(64, 2, 4)), strict=True)
    m = m.to(dev)
    m.compute_dtype = "bf16"
    opt = FusedAdam(m.parameters(), lr=1e-3, max_grad_norm=1.0)
    lrs = rand(1301, 1, 3, 3, 40, 72).to(dev)
    hr = rand(1302, 1, 3, 3, 160, 288).to(dev)
    try:
        # (a) quiet step: finite, parameters move
        before = opt.flat_params.clone()
        loss = VF.charbonnier_loss(m(lrs), hr)
        loss.backward()
        opt.step()
        assert bool(torch.isfinite(loss)) and bool(torch.isfinite(opt.last_grad_norm).all()) and not torch.equal(opt.flat_params, before)
        opt.zero_grad()
        # (b) the first forward chain launch gives up: sr and the loss show it
        before = opt.flat_params.clone()
        assert inject(1) == 0
        sr = m(lrs)
        loss = VF.charbonnier_loss(sr, hr)
        assert inject(0) == 0, "the forward did not launch a chain"
        assert not bool(torch.isfinite(sr).all()) and not bool(torch.isfinite(loss))
        loss.backward()
        opt.step()
        assert not bool(torch.isfinite(opt.last_grad_norm).all()) and torch.equal(opt.flat_params, before)
        opt.zero_grad()
        # (c) clean forward, the first BACKWARD chain launch gives up: the gradients show it, the step is skipped
        sr = m(lrs)
        loss = VF.charbonnier_loss(sr, hr)
        assert bool(torch.isfinite(sr).all())
        assert inject(1) == 0
        loss.backward()
        assert inject(0) == 0, "the backward did not launch a chain"
        opt.step()
        assert not bool(torch.isfinite(opt.last_grad_norm).all()) and torch.equal(opt.flat_params, before)
        assert opt.rewind_skipped_step() is True            # (no real timeout happened: nothing to raise)
        opt.zero_grad()
        # (d) and the step after that is clean again
        loss = VF.charbonnier_loss(m(lrs), hr)
        loss.backward()
        opt.step()
        assert bool(torch.isfinite(opt.last_grad_norm).all()) and not torch.equal(opt.flat_params, before)
    finally:
        inject(0)
        m._pool.clear()


_TIMEOUT_CHILD = r"""
import sys, torch
sys.path.insert(0, {root!r})
from oracle import basicvsr_oracle as O
from vsrlab_amd import functional as VF
from vsrlab_amd.vsr.models.RealBasicVSR.modules.basicvsr import BasicVSR
m = BasicVSR(64, 2, 4, False, False)
m.load_state_dict(O.keyed_state_dict(O.basicvsr_param_shapes(64, 2, 4)), strict=True)
m = m.to("cuda:0"); m.compute_dtype = "bf16"
g = torch.Generator().manual_seed(5)
lrs = torch.rand(1, 2, 3, 40, 72, generator=g).to("cuda:0")
sr = m(lrs)
loss = sr.mean()
torch.cuda.synchronize()
n = VF.chain_timeouts()
print("TIMEOUTS", n, "FINITE", bool(torch.isfinite(sr).all()), bool(torch.isfinite(loss)))
try:
    VF.raise_on_chain_timeout("child")
    print("RAISED 0")
except RuntimeError as e:
    print("RAISED 1", e)
"""


def test_a_real_chain_timeout_is_loud_in_a_subprocess():
    """The same failure through the REAL path: the diagnostic build `make ABL=8 ABLSRC=conv3x3_chain` (bit 3: the MFMA waves never
    publish their tiles) makes every dependency wait of a layer > 0 run into its 1 s clock.  A forward through that library must
    drain (no hang), count its give-ups, return a non-finite sr, and functional.raise_on_chain_timeout must raise."""
    _gpu()
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "vsrlab_amd", "lib", "libvsrlab_hip_conv3x3_chain_abl8.so")
    assert os.path.exists(lib), "build() makes the diagnostic chain library (make ABL=8 ABLSRC=conv3x3_chain)"
    env = dict(os.environ, VSRLAB_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-c", _TIMEOUT_CHILD.format(root=root)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("TIMEOUTS")][0].split()
    assert int(line[1]) > 0 and line[3] == "False" and line[4] == "False", r.stdout
    assert "RAISED 1" in r.stdout, r.stdout


def test_generic_kernel_switches_in_subprocesses():
    """VSRLAB_AMD_GENERIC_CONV=1 (an A/B switch read once per process: every 3x3 64 -> 64 convolution outside the trunk chains on the generic
    tiled kernel instead of the persistent one; the gradients into the pixel-shuffle layers stay in the strided layout).  Round 4 found it
    broken since round 2 -- the generic kernel writes no sign bits, conv_last.2's data gradient read garbage ones (gradient error 0.53) --
    and fixed it in Ctx::conv64; the whole-path goldens run again in a child with the switch on, also with one launch per layer."""
    _gpu()
    import subprocess
    import sys
    # (not the upscale-2 test: its diet-vs-full comparison to 1e-5 presumes the same kernels in both arenas, and with the switch on the diet
    # arena's per-layer backward runs on the generic kernel while the full arena's chains do not)
    sel = "test_basicvsr_end_to_end_vs_golden or test_basicvsr_train_flow_vs_golden or test_ragged_sizes_and_single_frame"
    # ... and the other process-wide implementation switches of INTEGRATION.md section 5 while a child is being paid for anyway
    for extra in ({"VSRLAB_AMD_GENERIC_CONV": "1"}, {"VSRLAB_AMD_GENERIC_CONV": "1", "VSRLAB_AMD_CHAIN": "0"}, {"VSRLAB_AMD_GENERIC_WGRAD": "1"},
                  {"VSRLAB_AMD_SINGLE_STREAM": "1", "VSRLAB_AMD_CHAIN": "0"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", sel],
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (extra, r.stdout[-2000:] + r.stderr[-2000:])


def test_parity_suite_on_the_diet_arena_in_a_subprocess():
    """The whole-path parity tests again with VSRLAB_AMD_ARENA=diet (goldens, oracle, noise-floor criteria unchanged): the switch
    is process-wide, so they run in a child."""
    _gpu()
    import subprocess
    import sys
    env = dict(os.environ, VSRLAB_AMD_ARENA="diet")
    sel = ("test_basicvsr_end_to_end_vs_golden or test_basicvsr_train_flow_vs_golden or test_basicvsr_input_gradient_vs_golden or "
           "test_ragged_sizes_and_single_frame or test_far_flows_match_the_oracle or test_realbasicvsr_training_vs_oracle or "
           "test_backward_is_linear_in_cotangent_large or (test_basicvsr_config1_all_grads_vs_oracle and bf16)")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", sel],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("shape", [(2, 24, 40), (3, 72, 200)])
def test_spynet_7x7_weight_gradients_on_the_producer_consumer_kernel(shape):
    """Weight / bias / input gradients of the five SPyNet layer shapes, bf16 build, through ConvReLU's autograd (vsr_conv_layer_bwd ->
    wgrad7x7_pc.hip: all seven kernel rows in one launch, 7 x G workgroups; the data gradient stays on the generic kernel): against
    torch autograd of the same convolution on the bf16-rounded operands (the engine also rounds the masked cotangent to bf16:
    1e-2 relative L2), from one tile per workgroup to several (3 x 72 x 200 = 189 tiles on 32 workgroups per kernel row: both
    buffer sets re-used), bit-identical on a repeat."""
    dev = _gpu()
    from vsrlab_amd import functional as VF
    n, h, w = shape
    for j, (ci, co) in enumerate([(8, 32), (32, 64), (64, 32), (32, 16), (16, 2)]):
        x0 = bf16_round(rand(500 + j, n, ci, h, w, lo=-1, hi=1)).to(dev)
        w0 = bf16_round(rand(510 + j, co, ci, 7, 7, lo=-1, hi=1) / (7.0 * ci ** 0.5)).to(dev)
        b0 = rand(520 + j, co, lo=-0.5, hi=0.5).to(dev)
        cot = rand(530 + j, n, co, h, w, lo=-1, hi=1).to(dev)

        def run(fn):
            x, wt, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            (fn(x, wt, b) * cot).sum().backward()
            return x.grad, wt.grad, b.grad

        got = run(lambda x, wt, b: VF.conv_relu_forward(x, wt, b, compute_dtype="bf16"))
        want = run(lambda x, wt, b: F.relu(F.conv2d(x, wt, b, padding=3)))
        for name, a, r in zip(("dx", "dw", "db"), got, want):
            assert rel_l2(a, r) < 1e-2, (shape, ci, co, name, rel_l2(a, r))
        again = run(lambda x, wt, b: VF.conv_relu_forward(x, wt, b, compute_dtype="bf16"))
        assert all(torch.equal(a, b) for a, b in zip(got, again)), (shape, ci, co)

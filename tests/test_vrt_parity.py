"""GPU parity tests of VRT's window attention (BASELINE config 5): the fused MFMA attention
(csrc/window_attention.hip) inside the WindowAttention / TMSA modules, against the reference's own fp64 outputs
(tests/golden/vrt_window_attention.npz) and the CPU oracle (oracle/vrt_attention_oracle.py).

fp32 build: 1e-3 relative; bf16 build: error <= 1.5 x the error of the oracle that rounds the qkv tensors and the
attention output to bf16 (the HIP build's storage points)."""
import ctypes

import pytest
import torch

from helpers import golden, proj_vector, rand, rel_err, rel_l2

pytestmark = pytest.mark.gpu

from oracle import basicvsr_oracle as O  # noqa: E402  (checker only)
from oracle import vrt_attention_oracle as V  # noqa: E402


def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests need an MI355X")
    return torch.device("cuda:0")


def _keyed(module):
    with torch.no_grad():
        for k, p in module.named_parameters():
            p.copy_(O.keyed_tensor(k, tuple(p.shape)))
    return module


CASES = {"a": (120, (2, 8, 8), True, 4), "b": (180, (6, 8, 8), False, 2)}


def _oracle_wa(tag, dtype64=True, emulate=False):
    dim, ws, mut, B_ = CASES[tag]
    from vsrlab_amd.vsr.models.VRT.modules.window_attention import WindowAttention
    m = _keyed(WindowAttention(dim, ws, 6, qkv_bias=True, mut_attn=mut))
    cast = (lambda t: t.double()) if dtype64 else (lambda t: t.float())
    sd = {k: (cast(v) if v.is_floating_point() else v) for k, v in m.state_dict().items()}
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and k != "position_bias" else v) for k, v in sd.items()}
    g = golden("vrt_window_attention")
    N = ws[0] * ws[1] * ws[2]
    x = cast(rand(int(g[f"{tag}__seed_x"]), B_, N, dim, lo=-1, hi=1)).requires_grad_(True)
    cot = cast(rand(int(g[f"{tag}__seed_cot"]), B_, N, dim, lo=-1, hi=1))
    mask = cast(V.compute_mask(2 * ws[0], 16, 16, ws, tuple(i // 2 for i in ws))[:2]) if tag == "a" else None
    if emulate:
        with O.emulate_bf16():
            y = V.window_attention_forward(leaves, x, mask, 6, mut)
            (y * cot).sum().backward()
    else:
        y = V.window_attention_forward(leaves, x, mask, 6, mut)
        (y * cot).sum().backward()
    grads = {k: v.grad for k, v in leaves.items() if v.is_floating_point() and v.requires_grad}
    grads["__dx"] = x.grad
    return y.detach(), grads


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_window_attention_module_vs_golden(tag, dtype):
    dev = _gpu()
    from vsrlab_amd.vsr.models.VRT.modules.window_attention import WindowAttention
    dim, ws, mut, B_ = CASES[tag]
    g = golden("vrt_window_attention")
    m = _keyed(WindowAttention(dim, ws, 6, qkv_bias=True, mut_attn=mut)).to(dev)
    m.compute_dtype = dtype
    N = ws[0] * ws[1] * ws[2]
    x = rand(int(g[f"{tag}__seed_x"]), B_, N, dim, lo=-1, hi=1).to(dev).requires_grad_(True)
    cot = rand(int(g[f"{tag}__seed_cot"]), B_, N, dim, lo=-1, hi=1).to(dev)
    mask = V.compute_mask(2 * ws[0], 16, 16, ws, tuple(i // 2 for i in ws))[:2].to(dev) if tag == "a" else None
    y = m(x, mask)
    (y * cot).sum().backward()
    got = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    got["__dx"] = x.grad.detach().cpu()
    y64, ref = _oracle_wa(tag)
    assert rel_err(y64, g[f"{tag}__out"]) < 1e-6                      # the oracle is the golden's equal (CPU test pins every tensor)
    assert set(got) == set(ref)
    if dtype == "fp32":
        assert rel_err(y, g[f"{tag}__out"]) < 1e-3 and rel_err(got["__dx"], g[f"{tag}__dx"]) < 1e-3
        for k in ref:
            assert rel_l2(got[k], ref[k]) < 1e-3, (k, rel_l2(got[k], ref[k]))
        for k, p in m.named_parameters():
            name = k.replace(".", "__")
            gn = float(g[f"{tag}__gnorm__{name}"])
            assert abs(float(p.grad.norm()) - gn) < 1e-3 * gn, k
            assert abs(float((p.grad.cpu().double() * proj_vector(k, tuple(p.shape))).sum()) - float(g[f"{tag}__gproj__{name}"])) < 1e-3 * gn * p.numel() ** 0.5, k
    else:
        y_e, emu = _oracle_wa(tag, dtype64=False, emulate=True)
        assert rel_err(y, y64) <= 1.5 * max(rel_err(y_e, y64), 1e-3), (rel_err(y, y64), rel_err(y_e, y64))
        from test_hip_parity import _noise_floor_check
        _noise_floor_check(got, emu, ref, max_glob_ratio=1.5, max_tensor_ratio=2.5)


def test_tmsa_block_vs_golden():
    """One shifted TMSA block (window (2,8,8), shift (1,4,4), H padded 20 -> 24), fp32 build: output, d/dx and all 17
    parameter gradients against the reference's fp64 run."""
    dev = _gpu()
    from vsrlab_amd.vsr.models.VRT.modules.tmsa import TMSA
    from vsrlab_amd.vsr.models.VRT.modules.window_attention import compute_mask
    g = golden("vrt_window_attention")
    blk = _keyed(TMSA(120, (4, 20, 16), 6, window_size=(2, 8, 8), shift_size=(1, 4, 4), mut_attn=True, mlp_ratio=2., qkv_bias=True)).to(dev)
    blk.attn.compute_dtype = "fp32"
    x = rand(70, 1, 4, 20, 16, 120, lo=-1, hi=1).to(dev).requires_grad_(True)
    cot = rand(71, 1, 4, 20, 16, 120, lo=-1, hi=1).to(dev)
    mask = compute_mask(4, 24, 16, (2, 8, 8), (1, 4, 4), dev)
    assert torch.equal(mask.cpu(), g["t__mask"])
    y = blk(x, mask)
    (y * cot).sum().backward()
    assert rel_err(y, g["t__out"]) < 1e-3 and rel_err(x.grad, g["t__dx"]) < 1e-3
    n = 0
    for k, p in blk.named_parameters():
        name = k.replace(".", "__")
        if f"t__grad__{name}" in g:
            assert rel_l2(p.grad, g[f"t__grad__{name}"]) < 1e-3, k
        gn = float(g[f"t__gnorm__{name}"])
        assert abs(float(p.grad.norm()) - gn) < 1e-3 * gn, k
        assert abs(float((p.grad.cpu().double() * proj_vector(k, tuple(p.shape))).sum()) - float(g[f"t__gproj__{name}"])) < 1e-3 * gn * p.numel() ** 0.5, k
        n += 1
    assert n == 17


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_tmsag_and_rtmsa_vs_golden(dtype):
    """The two containers around TMSA (tmsa.py:126-251) with the HIP attention inside: TMSAG depth 3 with mutual attention on a padded
    volume (block 1 shifted, one mask per call) and RTMSA depth 2 on (6,8,8) windows (N = 384 keys in registers).  fp32 build: output,
    d/dx and every parameter gradient (norm + seeded projection) against the reference's fp64 run at 1e-3; bf16 build: output and
    d/dx error <= 1.5 x the error of the bf16-emulating oracle."""
    dev = _gpu()
    from vsrlab_amd.vsr.models.VRT.modules.tmsa import RTMSA, TMSAG
    g = golden("vrt_groups")
    cases = (("g", lambda: TMSAG(120, (4, 20, 16), 3, 6, window_size=[2, 8, 8], mut_attn=True, mlp_ratio=2., qkv_bias=True), (1, 120, 4, 20, 16),
              lambda sd, x: V.tmsag_forward(sd, x, 6, (2, 8, 8), None, True, 3)),
             ("r", lambda: RTMSA(180, (6, 16, 16), 2, 6, window_size=[6, 8, 8], mlp_ratio=2., qkv_bias=True), (1, 180, 6, 16, 16),
              lambda sd, x: V.rtmsa_forward(sd, x, 6, (6, 8, 8), 2)))
    for tag, make, shape, fwd in cases:
        m = _keyed(make()).to(dev)
        for mod in m.modules():
            if hasattr(mod, "compute_dtype"):
                mod.compute_dtype = dtype
        x = rand(int(g[f"{tag}__seed_x"]), *shape, lo=-1, hi=1).to(dev).requires_grad_(True)
        cot = rand(int(g[f"{tag}__seed_cot"]), *shape, lo=-1, hi=1).to(dev)
        y = m(x)
        (y * cot).sum().backward()
        if dtype == "fp32":
            assert rel_err(y, g[f"{tag}__out"]) < 1e-3
            grads = {k: p.grad for k, p in m.named_parameters()}
            grads["dx"] = x.grad
            for k, gr in grads.items():
                name = k.replace(".", "__")
                gn = float(g[f"{tag}__gnorm__{name}"])
                assert abs(float(gr.norm()) - gn) < 1e-3 * gn, (tag, k)
                assert abs(float((gr.cpu().double() * proj_vector(k, tuple(gr.shape))).sum()) - float(g[f"{tag}__gproj__{name}"])) < 1e-3 * gn * gr.numel() ** 0.5, (tag, k)
        else:
            cpu = make()
            _keyed(cpu)
            named = dict(cpu.named_parameters())
            sd = {k: (v.detach().clone() if k in named else v) for k, v in cpu.state_dict().items()}
            x64 = rand(int(g[f"{tag}__seed_x"]), *shape, lo=-1, hi=1).double().requires_grad_(True)
            y64 = fwd({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, x64)
            (y64 * cot.cpu().double()).sum().backward()
            assert rel_err(y64, g[f"{tag}__out"]) < 1e-6
            xe = rand(int(g[f"{tag}__seed_x"]), *shape, lo=-1, hi=1).requires_grad_(True)
            with O.emulate_bf16():
                ye = fwd(sd, xe)
                (ye * cot.cpu()).sum().backward()
            assert rel_err(y, y64) <= 1.5 * max(rel_err(ye, y64), 1e-3), (tag, rel_err(y, y64), rel_err(ye, y64))
            assert rel_err(x.grad, x64.grad) <= 1.5 * max(rel_err(xe.grad, x64.grad), 1e-3), (tag, rel_err(x.grad, x64.grad), rel_err(xe.grad, x64.grad))


def test_window_attention_config5_scale_properties():
    """BASELINE config 5's stage-1 shape through the C ABI (bf16): 320x180 LR padded to 320x184, 16 frames, window (2,8,8)
    -> 7360 windows x 128 tokens, dim 120, 6 heads.  Size-independent properties: with v == 1 every output is 1 (softmax
    rows sum to one, masked and biased or not); finite gradients; d/dv of sum(out) sums to the number of queries per key
    set; a timing line for the fused kernel."""
    dev = _gpu()
    import vsrlab_amd
    from vsrlab_amd import functional as VF
    from vsrlab_amd._lib import AttnDesc
    lib = vsrlab_amd._lib.load()
    B, N, heads, hd = 8 * 23 * 40, 128, 6, 20
    C = heads * hd
    qkv = torch.randn(B, N, 3, heads, hd, device=dev)
    qkv[:, :, 2] = 1.0
    qkv = qkv.to(torch.bfloat16).contiguous()
    bias = torch.randn(heads, N, N, device=dev)
    mask = V.compute_mask(16, 184, 320, (2, 8, 8), (1, 4, 4)).to(dev)
    nW = mask.shape[0]
    assert B % nW == 0
    out = torch.empty(B, N, C, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, heads, N, device=dev)
    d = AttnDesc(B, N, heads, hd, 0, 0, 0, N, N, C, 0, nW, N, hd ** -0.5, VF.DT_BF16)
    st = VF._stream()
    assert lib.vsr_window_attention_fwd(ctypes.byref(d), VF._ptr(qkv), VF._ptr(bias), VF._ptr(mask), VF._ptr(out), VF._ptr(lse), st) == 0
    torch.cuda.synchronize()
    assert float((out.float() - 1).abs().max()) < 1e-2
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.vsr_window_attention_fwd(ctypes.byref(d), VF._ptr(qkv), VF._ptr(bias), VF._ptr(mask), VF._ptr(out), VF._ptr(lse), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    flops = 4.0 * B * heads * N * N * hd
    print(f"window attention fwd, config-5 stage shape: {ms:.3f} ms, {flops / ms / 1e9:.1f} TFLOP/s useful (head_dim 20 of a 32-deep MFMA step)")
    dout = torch.ones_like(out)
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty_like(lse)
    dbias = torch.zeros_like(bias)
    assert lib.vsr_window_attention_bwd(ctypes.byref(d), VF._ptr(qkv), VF._ptr(bias), VF._ptr(mask), VF._ptr(dout), VF._ptr(lse), VF._ptr(delta),
                                        VF._ptr(dqkv), VF._ptr(dbias), st) == 0
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dqkv.float()).all()) and bool(torch.isfinite(dbias).all())
    # out = P v: d sum(out) / d v[key, c] = sum_queries P[query, key]; summed over keys = number of queries
    dv = dqkv[:, :, 2].float()
    assert abs(float(dv[0, :, 0, 0].sum()) - N) < 0.05 * N
    # v is constant, so the scores do not influence the output: dq, dk and dbias vanish up to rounding
    assert float(dqkv[:, :, 0].float().abs().max()) < 5e-2 and float(dbias.abs().max()) < 0.5 * B / 100


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_vrt_spynet_multi_level_vs_golden(dtype):
    """vsrlab.vsr.models.VRT.modules.spynet.SpyNet (conf/train/model/spynet.yaml: return_levels [2,3,4,5]) on the HIP SPyNet
    engine without the final ReLU: the four flows against the reference's own outputs."""
    dev = _gpu()
    from vsrlab_amd.vsr.models.VRT.modules.spynet import SpyNet
    g = golden("vrt_spynet")
    m = SpyNet(False, [2, 3, 4, 5])
    sd = {k: (v if k in ("mean", "std") else O.keyed_tensor(k, tuple(v.shape)) * (3.0 if k.endswith("weight") else 1.0))
          for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    import os
    os.environ["VSRLAB_AMD_DTYPE"] = dtype
    try:
        for tag, shape in (("a", (1, 3, 64, 96)), ("b", (2, 3, 40, 72))):
            ref, supp = rand(int(g[f"{tag}__seed_ref"]), *shape).to(dev), rand(int(g[f"{tag}__seed_supp"]), *shape).to(dev)
            with torch.no_grad():
                flows = m(ref, supp)
            assert len(flows) == 4
            for i, f in enumerate(flows):
                assert rel_err(f, g[f"{tag}__flow{i}"]) < (1e-3 if dtype == "fp32" else 6e-2), (tag, i, rel_err(f, g[f"{tag}__flow{i}"]))
    finally:
        del os.environ["VSRLAB_AMD_DTYPE"]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_vrt_spynet_multi_level_gradients_vs_oracle(dtype):
    """The same module under autograd (vsr/models/VRT/modules/spynet.py:98-157 is an ordinary differentiable module): the loss
    sum_i mean(flow_i * cot_i) over the four returned levels, gradients of the 60 conv tensors and of both frames against
    torch autograd on the oracle's restatement in fp64 (pinned to the reference's forward by vrt_spynet.npz; the reference
    holds no gradient fixture for this module, and the BasicVSR-path SPyNet backward the engine shares is pinned by
    basicvsr_m64_rb3_trainflow.npz).  Criterion: the coarse-to-fine recursion amplifies rounding into ReLU-mask flips, so the
    HIP build's error against fp64 is bounded by 2 x the error of the SAME-precision CPU evaluation (fp32 oracle / oracle with
    bf16 storage emulated), floors 2e-3 / 2e-2, globally and per tensor (3 x); a /32 size and one that is resized."""
    dev = _gpu()
    from vsrlab_amd.vsr.models.VRT.modules.spynet import SpyNet
    levels = [2, 3, 4, 5]
    m = SpyNet(False, levels)
    sd = {k: (v if k in ("mean", "std") else O.keyed_tensor(k, tuple(v.shape)) * (3.0 if k.endswith("weight") else 1.0))
          for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    import os
    os.environ["VSRLAB_AMD_DTYPE"] = dtype
    try:
        for shape in ((1, 3, 64, 96), (2, 3, 40, 72)):
            n, _, h, w = shape
            ref0, supp0 = rand(401, *shape), rand(402, *shape)
            cots = [rand(410 + lv, n, 2, h >> (5 - lv), w >> (5 - lv), lo=-1, hi=1) for lv in sorted(levels, reverse=True)]

            def oracle_grads(cast, emulate):
                leaves = {k: (cast(v).requires_grad_(True) if k not in ("mean", "std") else cast(v)) for k, v in sd.items()}
                r, s_ = cast(ref0).requires_grad_(True), cast(supp0).requires_grad_(True)
                if emulate:
                    with O.emulate_bf16():
                        flows = V.vrt_spynet_forward(leaves, r, s_, tuple(levels))
                        sum((f * cast(c)).mean() for f, c in zip(flows, cots)).backward()
                else:
                    flows = V.vrt_spynet_forward(leaves, r, s_, tuple(levels))
                    sum((f * cast(c)).mean() for f, c in zip(flows, cots)).backward()
                out = {k: v.grad.double() for k, v in leaves.items() if k not in ("mean", "std")}
                out["ref"], out["supp"] = r.grad.double(), s_.grad.double()
                return out

            want = oracle_grads(lambda t: t.detach().clone().double(), False)
            same = oracle_grads(lambda t: t.detach().clone().float(), dtype == "bf16")
            for p in m.parameters():
                p.grad = None
            rg, sg = ref0.clone().to(dev).requires_grad_(True), supp0.clone().to(dev).requires_grad_(True)
            flows = m(rg, sg)
            sum((f * c.to(dev)).mean() for f, c in zip(flows, cots)).backward()
            got = {k: p.grad.double().cpu() for k, p in m.named_parameters()}
            got["ref"], got["supp"] = rg.grad.double().cpu(), sg.grad.double().cpu()
            assert set(got) == set(want) and len(got) == 62
            floor = 2e-3 if dtype == "fp32" else 2e-2
            keys = sorted(want)
            cat = lambda d: torch.cat([d[k].flatten() for k in keys])
            e_hip, e_same = rel_l2(cat(got), cat(want)), rel_l2(cat(same), cat(want))
            assert e_hip <= 2.0 * max(e_same, floor), (shape, e_hip, e_same)
            for k in keys:
                a, b = rel_l2(got[k], want[k]), rel_l2(same[k], want[k])
                assert a <= 3.0 * max(b, floor), (shape, k, a, b)
    finally:
        del os.environ["VSRLAB_AMD_DTYPE"]

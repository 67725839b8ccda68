"""GPU tests of the training-step glue (SURVEY.md 8f rank 3 + the loss term of rank 1): fused clip + Adam over flat
arenas, the gradient arena written directly by the HIP backward, compute_loss's resize term.  Checkers: torch's own
CPU implementations of the reference's calls (torch.optim.Adam, clip_grad_norm_, F.interpolate) -- what
core/utils.py:235-240,270-280 invoke -- and the CPU oracle for the model."""
import pytest
import torch
import torch.nn.functional as F

from helpers import rand, rel_err, rel_l2

pytestmark = pytest.mark.gpu

from oracle import basicvsr_oracle as O  # noqa: E402  (checker only)


def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests need an MI355X")
    return torch.device("cuda:0")


SHAPES = [(64, 67, 3, 3), (64,), (3, 64, 3, 3), (3,), (64, 128, 1, 1), (7, 5), (1,), (256, 64, 3, 3)]


@pytest.mark.parametrize("max_norm", [None, 1.0, 1e-2])
def test_fused_adam_matches_torch_adam_and_clip_grad_norm(max_norm):
    """5 steps on 8 tensors of ragged sizes with the reference's hyper-parameters (conf/train/optimizer/adam.yaml:
    lr 1e-4, betas (0.9, 0.99), eps 1e-8) and gradient_clip_val (conf/train/default.yaml:20) against
    torch.nn.utils.clip_grad_norm_ + torch.optim.Adam on the CPU.  Tolerance: fp32 rounding (the reduction order of the
    norm and fused multiply-adds differ), 2e-6 relative on the parameters."""
    dev = _gpu()
    from vsrlab_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(5)
    ref = [torch.nn.Parameter(torch.randn(*s, generator=g) * 0.05) for s in SHAPES]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]
    opt_r = torch.optim.Adam(ref, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0)
    opt_m = FusedAdam(mine, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0, max_grad_norm=max_norm)
    for step in range(5):
        scale = 10.0 if step == 3 else 0.1              # one step far above the clip threshold
        grads = [torch.randn(*s, generator=g) * scale for s in SHAPES]
        for p, q, gr in zip(ref, mine, grads):
            p.grad = gr.clone()
            q.grad.copy_(gr.to(dev))                      # the arena view stays attached
        tn = torch.nn.utils.clip_grad_norm_(ref, max_norm if max_norm is not None else 1e30)
        opt_r.step()
        opt_m.step()
        assert abs(float(opt_m.last_grad_norm.item()) - float(tn)) < 2e-6 * float(tn)
        for p, q in zip(ref, mine):
            assert rel_err(q.detach().cpu(), p.detach()) < 2e-6
        opt_r.zero_grad()
        opt_m.zero_grad()
        assert float(opt_m.flat_grads.abs().max()) == 0.0 and all(q.grad is not None for q in mine)
    for p, q in zip(ref, mine):
        assert rel_err(opt_m.state[q]["exp_avg"].cpu(), opt_r.state[p]["exp_avg"]) < 1e-5
        assert rel_err(opt_m.state[q]["exp_avg_sq"].cpu(), opt_r.state[p]["exp_avg_sq"]) < 1e-5
    # checkpoint interchange: torch.optim.Adam's state_dict loads into FusedAdam and continues identically
    mine2 = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]
    opt2 = FusedAdam(mine2, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, max_grad_norm=max_norm)
    opt2.load_state_dict(opt_r.state_dict())
    grads = [torch.randn(*s, generator=g) for s in SHAPES]
    for p, q, gr in zip(ref, mine2, grads):
        p.grad = gr.clone()
        q.grad.copy_(gr.to(dev))
    torch.nn.utils.clip_grad_norm_(ref, max_norm if max_norm is not None else 1e30)
    opt_r.step()
    opt2.step()
    for p, q in zip(ref, mine2):
        assert rel_err(q.detach().cpu(), p.detach()) < 2e-6
    assert set(opt2.state_dict()["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}


def test_fused_adam_skips_the_step_on_non_finite_gradients_and_weight_decay():
    dev = _gpu()
    from vsrlab_amd.optim import FusedAdam
    q = torch.nn.Parameter(torch.ones(1000, device=dev))
    opt = FusedAdam([q], lr=1e-2, weight_decay=0.1)
    q.grad.fill_(float("nan"))
    opt.step()
    assert torch.equal(q.detach().cpu(), torch.ones(1000))             # GradScaler.step semantics on inf / nan
    p = torch.nn.Parameter(torch.ones(1000))
    ref = torch.optim.Adam([p], lr=1e-2, weight_decay=0.1, betas=(0.9, 0.99))
    q2 = torch.nn.Parameter(torch.ones(1000, device=dev))
    opt2 = FusedAdam([q2], lr=1e-2, weight_decay=0.1)
    for _ in range(3):
        p.grad = torch.full((1000,), 0.3)
        q2.grad.fill_(0.3)
        ref.step(); opt2.step()
    assert rel_err(q2.detach().cpu(), p.detach()) < 2e-6


@pytest.mark.gpu
def test_fused_adam_skipped_step_count():
    """round-2 ADVICE: the host-side step count advances on a device-skipped update (documented in FusedAdam.step);
    rewind_skipped_step() restores what GradScaler.step + torch.optim.Adam would hold."""
    dev = _gpu()
    from vsrlab_amd.optim import FusedAdam
    q = torch.nn.Parameter(torch.ones(64, device=dev))
    opt = FusedAdam([q], lr=1e-2)
    q.grad.fill_(0.5)
    opt.step()
    assert int(opt.state_dict()["state"][0]["step"]) == 1 and not opt.rewind_skipped_step()
    q.grad.fill_(float("inf"))
    before = q.detach().clone()
    opt.step()
    assert torch.equal(q.detach(), before)
    assert int(opt.state_dict()["state"][0]["step"]) == 2          # the documented deviation
    assert opt.rewind_skipped_step()
    assert int(opt.state_dict()["state"][0]["step"]) == 1          # torch's count after a step GradScaler would have skipped
    # the next real step then has torch's bias corrections
    p = torch.nn.Parameter(torch.ones(64))
    ref = torch.optim.Adam([p], lr=1e-2, betas=(0.9, 0.99))
    for g in (0.5, 0.25):
        p.grad = torch.full((64,), g)
        ref.step()
    q.grad.fill_(0.25)
    opt.step()
    assert rel_err(q.detach().cpu(), p.detach()) < 2e-6


@pytest.mark.parametrize("shape,size", [((2, 3, 3, 64, 96), (16, 24)), ((1, 2, 3, 37, 53), (9, 13)), ((2, 3, 20, 28), (40, 56))])
def test_resize_matches_interpolate(shape, size):
    """kornia.geometry.transform.resize(hr, (h, w)) = F.interpolate(bilinear, align_corners=False) on (..., H, W)."""
    dev = _gpu()
    from vsrlab_amd.core.utils import resize
    x = rand(3, *shape)
    want = F.interpolate(x.reshape(-1, 1, *shape[-2:]), size=size, mode="bilinear", align_corners=False).reshape(*shape[:-2], *size)
    got = resize(x.to(dev), size).cpu()
    assert got.shape == want.shape and rel_err(got, want) < 1e-6


def test_compute_loss_two_terms_value_and_gradients():
    """compute_loss(loss_fn, sr, hr, lq) (core/utils.py:235-240) with CharbonnierLoss: value and d/dsr, d/dlq."""
    dev = _gpu()
    from vsrlab_amd.core.losses import CharbonnierLoss
    from vsrlab_amd.core.utils import compute_loss
    sr = rand(1, 1, 2, 3, 64, 96).requires_grad_(True)
    hr = rand(2, 1, 2, 3, 64, 96)
    lq = rand(3, 1, 2, 3, 16, 24).requires_grad_(True)
    tgt = F.interpolate(hr.reshape(-1, 3, 64, 96), size=(16, 24), mode="bilinear", align_corners=False).reshape(1, 2, 3, 16, 24)
    want = O.charbonnier(sr, hr) + O.charbonnier(lq, tgt)
    want.backward()
    sr_d = sr.detach().to(dev).requires_grad_(True)
    lq_d = lq.detach().to(dev).requires_grad_(True)
    got = compute_loss(CharbonnierLoss(), sr_d, hr.to(dev), lq_d)
    got.backward()
    assert abs(float(got) - float(want)) < 1e-5 * float(want)
    assert rel_err(sr_d.grad.cpu(), sr.grad) < 1e-5 and rel_err(lq_d.grad.cpu(), lq.grad) < 1e-5
    only = compute_loss(CharbonnierLoss(), sr_d.detach(), hr.to(dev))
    assert abs(float(only) - float(O.charbonnier(sr.detach(), hr))) < 1e-5 * float(only)
    # a contiguous fp32 view at a storage offset that is not a multiple of 4 elements (round-3 ADVICE: the float4 kernel refused it;
    # the reference loss accepts any tensor): sr[0, 1:] of odd-sized planes
    a = rand(4, 2, 3, 5, 7)
    b = rand(5, 2, 3, 5, 7)
    av = a.to(dev)[0, 1:].requires_grad_(True)
    assert av.is_contiguous() and av.data_ptr() % 16 != 0
    got = CharbonnierLoss()(av, b.to(dev)[0, 1:])
    got.backward()
    ac = a[0, 1:].clone().requires_grad_(True)
    want = O.charbonnier(ac, b[0, 1:])
    want.backward()
    assert abs(float(got) - float(want)) < 1e-5 * float(want) and rel_err(av.grad.cpu(), ac.grad) < 1e-5


def test_backward_writes_the_optimizer_arena_directly():
    """With FusedAdam the HIP backward accumulates into the flat gradient arena (p.grad views): same values as the
    autograd path (bitwise at t=1: no atomics), += across micro-steps, survives zero_grad(set_to_none=True), and a
    second backward through the same graph raises instead of reading a released workspace."""
    dev = _gpu()
    from vsrlab_amd.optim import FusedAdam
    from vsrlab_amd.vsr.models.RealBasicVSR.modules.basicvsr import BasicVSR
    sd = O.keyed_state_dict(O.basicvsr_param_shapes(64, 2, 4))

    def make():
        m = BasicVSR(64, 2, 4, False, False)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev)
        m.compute_dtype = "fp32"
        return m

    lrs = rand(81, 1, 1, 3, 24, 40).to(dev)
    cot = rand(82, 1, 1, 3, 96, 160, lo=-1, hi=1).to(dev)
    a = make()
    torch.mean(a(lrs) * cot).backward()
    plain = {k: p.grad.clone() for k, p in a.named_parameters() if p.grad is not None}
    b = make()
    opt = FusedAdam(b.parameters(), lr=1e-4, max_grad_norm=1.0)
    assert len(opt._params) == len(plain)
    w_before = {k: p.detach().clone() for k, p in b.named_parameters()}
    sr = b(lrs)
    loss = torch.mean(sr * cot)
    loss.backward()
    for k, p in b.named_parameters():
        if p.requires_grad:
            assert p.grad.data_ptr() == p._vsr_grad_slot.data_ptr()
            assert torch.equal(p.grad, plain[k]), k
    with pytest.raises(RuntimeError):
        loss.backward()                                               # graph consumed: loud, not garbage
    b.zero_grad(set_to_none=True)                                     # a caller that drops .grad ...
    opt.zero_grad()
    torch.mean(b(lrs) * cot).backward()
    torch.mean(b(lrs) * cot).backward()                               # ... and accumulates two micro-steps
    for k, p in b.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and rel_err(p.grad, 2 * plain[k]) < 1e-6, k
    opt.step()
    ref = [torch.nn.Parameter(w_before[k].cpu()) for k, p in b.named_parameters() if p.requires_grad]
    for q, (k, p) in zip(ref, [(k, p) for k, p in b.named_parameters() if p.requires_grad]):
        q.grad = 2 * plain[k].cpu()
    torch.nn.utils.clip_grad_norm_(ref, 1.0)
    torch.optim.Adam(ref, lr=1e-4, betas=(0.9, 0.99)).step()
    for q, (k, p) in zip(ref, [(k, p) for k, p in b.named_parameters() if p.requires_grad]):
        assert rel_err(p.detach().cpu(), q.detach()) < 2e-6, k
    assert all(not p.requires_grad and p.grad is None for k, p in b.named_parameters() if "spynet" in k)

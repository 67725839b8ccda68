"""Shared helpers for the test-suite (seeded inputs identical to tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rand(seed, *shape, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(int(seed))
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files if z[k].dtype.kind in "fiub"}   # string arrays (key lists) are read by the tests that need them


def rel_err(a, b):
    """max|a-b| / max|b|  -- the 'relative fp32' measure used by every parity test."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def proj_vector(key, shape):
    """Seeded random direction for a parameter (same function as tests/golden/make_golden.py): gradients too
    large to store are pinned by <grad, proj_vector>, their sum and their L2 norm."""
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(("proj:" + key).encode()) & 0x7fffffff)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


def realbasicvsr_shapes(mid, res_blocks, cleaning_blocks):
    """state_dict key -> shape of RealBasicVSR(cleaning_blocks, mid_channels=mid, res_blocks=res_blocks)."""
    from oracle import basicvsr_oracle as O
    shapes = {"basicvsr." + k: s for k, s in O.basicvsr_param_shapes(mid, res_blocks, 4).items()}
    shapes.update(O.cleaner_param_shapes(mid, cleaning_blocks))
    return shapes


def realbasicvsr_oracle_grads(sd64, lr, cot_sr, cot_lq):
    """fp64 oracle: d [mean(sr*cot_sr) + mean(lq*cot_lq)] / d every trainable tensor (frozen SPyNet)."""
    from oracle import basicvsr_oracle as O
    leaves = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "spynet" not in k) for k, v in sd64.items()}
    sr, lq = O.realbasicvsr_forward(leaves, lr)
    (torch.mean(sr * cot_sr) + torch.mean(lq * cot_lq)).backward()
    return sr.detach(), lq.detach(), {k: v.grad for k, v in leaves.items() if v.grad is not None}

"""CPU restatement of VRT's window attention and TMSA block (BASELINE config 5).

TEST INFRASTRUCTURE ONLY.  Functional, state_dict-keyed; every function cites the reference lines it restates (paths
relative to /root/reference/src).  Pinned by tests/golden/vrt_window_attention.npz, generated from the imported
reference by tests/golden/make_golden.py (`vrt`)."""
from __future__ import annotations

from functools import reduce
from operator import mul
from typing import Mapping, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from .basicvsr_oracle import _q      # identity unless basicvsr_oracle.emulate_bf16() is active: rounds a stored tensor (and its gradient)


def window_partition(x: Tensor, ws) -> Tensor:
    """vsr/models/VRT/modules/window_attention.py:9-25."""
    B, D, H, W, C = x.shape
    x = x.view(B, D // ws[0], ws[0], H // ws[1], ws[1], W // ws[2], ws[2], C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, reduce(mul, ws), C)


def window_reverse(windows: Tensor, ws, B, D, H, W) -> Tensor:
    """window_attention.py:27-43."""
    x = windows.view(B, D // ws[0], H // ws[1], W // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().view(B, D, H, W, -1)


def compute_mask(D, H, W, ws, shift) -> Tensor:
    """window_attention.py:61-77: region labels of the cyclically shifted volume -> 0 / -100 per window."""
    img = torch.zeros((1, D, H, W, 1))
    cnt = 0
    for d in slice(-ws[0]), slice(-ws[0], -shift[0]), slice(-shift[0], None):
        for h in slice(-ws[1]), slice(-ws[1], -shift[1]), slice(-shift[1], None):
            for w in slice(-ws[2]), slice(-ws[2], -shift[2]), slice(-shift[2], None):
                img[:, d, h, w, :] = cnt
                cnt += 1
    mw = window_partition(img, ws).squeeze(-1)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def _attention(q, k, v, mask, scale, bias):
    """WindowAttention.attention (window_attention.py:140-162).  q, k, v: (B_, nH, N, hd); bias (nH, N, N) or None;
    mask (nW, Nm, Nm) or None: its top-left N x N block is added per window (B_ // nW windows share a mask row)."""
    B_, nH, N, hd = q.shape
    attn = (q * scale) @ k.transpose(-2, -1)
    if bias is not None:
        attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, nH, N, N) + mask[:, :N, :N].unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, nH, N, N)
    attn = attn.softmax(dim=-1)
    return (attn @ v).transpose(1, 2).reshape(B_, N, nH * hd)


def window_attention_forward(sd: Mapping[str, Tensor], x: Tensor, mask: Optional[Tensor], heads: int, mut_attn: bool,
                             prefix: str = "") -> Tensor:
    """WindowAttention.forward (window_attention.py:116-138).  x: (B_, N, C)."""
    B_, N, C = x.shape
    hd = C // heads
    scale = hd ** -0.5
    g = lambda k: sd[prefix + k]
    # _q: the HIP bf16 build stores the qkv tensors and the attention output in bf16 (the Linears themselves are library GEMMs)
    qkv = _q(F.linear(x, g("qkv_self.weight"), sd.get(prefix + "qkv_self.bias"))).reshape(B_, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    idx = g("relative_position_index")[:N, :N].reshape(-1)
    bias = g("relative_position_bias_table")[idx].reshape(N, N, -1).permute(2, 0, 1)           # :146-148
    x_out = _attention(qkv[0], qkv[1], qkv[2], mask, scale, bias)
    if mut_attn:
        qkv = _q(F.linear(x + g("position_bias").repeat(1, 2, 1), g("qkv_mut.weight"), sd.get(prefix + "qkv_mut.bias")))
        qkv = qkv.reshape(B_, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
        (q1, q2), (k1, k2), (v1, v2) = torch.chunk(qkv[0], 2, dim=2), torch.chunk(qkv[1], 2, dim=2), torch.chunk(qkv[2], 2, dim=2)
        x1 = _attention(q2, k1, v1, mask, scale, None)          # :128-131
        x2 = _attention(q1, k2, v2, mask, scale, None)
        x_out = torch.cat([torch.cat([x1, x2], 1), x_out], 2)
    return F.linear(_q(x_out), g("proj.weight"), g("proj.bias"))


def tmsa_forward(sd: Mapping[str, Tensor], x: Tensor, mask_matrix: Optional[Tensor], heads: int, window_size, shift_size,
                 mut_attn: bool) -> Tensor:
    """TMSA.forward (vsr/models/VRT/modules/tmsa.py:60-124), drop_path = 0.  x: (B, D, H, W, C)."""
    B, D, H, W, C = x.shape
    ws, ss = list(window_size), list(shift_size)
    for i, s in enumerate((D, H, W)):                              # get_window_size, window_attention.py:45-59
        if s <= ws[i]:
            ws[i], ss[i] = s, 0
    y = F.layer_norm(x, (C,), sd["norm1.weight"], sd["norm1.bias"])
    pd, pb, pr = (ws[0] - D % ws[0]) % ws[0], (ws[1] - H % ws[1]) % ws[1], (ws[2] - W % ws[2]) % ws[2]
    y = F.pad(y, (0, 0, 0, pr, 0, pb, 0, pd))
    _, Dp, Hp, Wp, _ = y.shape
    shifted = any(s > 0 for s in ss)
    if shifted:
        y = torch.roll(y, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
    win = window_partition(y, ws)
    att = window_attention_forward(sd, win, mask_matrix if shifted else None, heads, mut_attn, prefix="attn.")
    y = window_reverse(att.view(-1, *(tuple(ws) + (C,))), ws, B, Dp, Hp, Wp)
    if shifted:
        y = torch.roll(y, shifts=(ss[0], ss[1], ss[2]), dims=(1, 2, 3))
    y = y[:, :D, :H, :W, :]
    x = x + y
    z = F.layer_norm(x, (C,), sd["norm2.weight"], sd["norm2.bias"])
    z = F.linear(F.gelu(F.linear(z, sd["mlp.fc11.weight"], sd["mlp.fc11.bias"])) * F.linear(z, sd["mlp.fc12.weight"], sd["mlp.fc12.bias"]),
                 sd["mlp.fc2.weight"], sd["mlp.fc2.bias"])          # Mlp_GEGLU, window_attention.py:79-98
    return x + z


def tmsag_forward(sd: Mapping[str, Tensor], x: Tensor, heads: int, window_size, shift_size, mut_attn: bool, depth: int) -> Tensor:
    """TMSAG.forward (vsr/models/VRT/modules/tmsa.py:177-202): depth TMSA blocks, block i shifted iff i is odd (:168), one mask
    computed per call from the padded volume (:186-191).  x: (B, C, D, H, W); keys blocks.{i}.*"""
    B, C, D, H, W = x.shape
    ws, ss = list(window_size), list(shift_size if shift_size is not None else [w // 2 for w in window_size])
    full_ss = list(ss)
    for i, s in enumerate((D, H, W)):                              # get_window_size on the GROUP's sizes (:185)
        if s <= ws[i]:
            ws[i], ss[i] = s, 0
    Dp, Hp, Wp = -(-D // ws[0]) * ws[0], -(-H // ws[1]) * ws[1], -(-W // ws[2]) * ws[2]
    mask = compute_mask(Dp, Hp, Wp, ws, ss).to(x.dtype)
    y = x.permute(0, 2, 3, 4, 1)
    for i in range(depth):
        blk = {k[len(f"blocks.{i}."):]: v for k, v in sd.items() if k.startswith(f"blocks.{i}.")}
        y = tmsa_forward(blk, y, mask, heads, tuple(window_size), (0, 0, 0) if i % 2 == 0 else tuple(full_ss), mut_attn)
    return y.reshape(B, D, H, W, -1).permute(0, 4, 1, 2, 3)


def rtmsa_forward(sd: Mapping[str, Tensor], x: Tensor, heads: int, window_size, depth: int) -> Tensor:
    """RTMSA.forward (tmsa.py:250-251): x + linear(TMSAG(x) channels-last), the group without mutual attention (:236)."""
    grp = {k[len("residual_group."):]: v for k, v in sd.items() if k.startswith("residual_group.")}
    y = tmsag_forward(grp, x, heads, window_size, None, False, depth)
    return x + F.linear(y.transpose(1, 4), sd["linear.weight"], sd["linear.bias"]).transpose(1, 4)


def vrt_spynet_forward(sd: Mapping[str, Tensor], ref: Tensor, supp: Tensor, return_levels=(5,)):
    """SpyNet.forward / .process of the VRT tree (vsr/models/VRT/modules/spynet.py:98-157): the canonical SPyNet -- no ReLU
    after a level's last conv (:68-82), keys basic_module.{l}.basic_module.{0,2,4,6,8} -- returning the flows of the
    requested pyramid levels, finest first, each resized to (h // scale, w // scale) and rescaled (:134-141)."""
    from .basicvsr_oracle import flow_warp, _q, _wq      # _q / _wq: identity unless the oracle emulates bf16 storage (emulate_bf16)
    import math
    h, w = ref.shape[2:]
    w_floor = math.floor(math.ceil(w / 32.0) * 32.0)
    h_floor = math.floor(math.ceil(h / 32.0) * 32.0)
    ref = F.interpolate(ref, size=(h_floor, w_floor), mode="bilinear", align_corners=False)
    supp = F.interpolate(supp, size=(h_floor, w_floor), mode="bilinear", align_corners=False)
    refs, supps = [(ref - sd["mean"]) / sd["std"]], [(supp - sd["mean"]) / sd["std"]]
    for _ in range(5):
        refs.insert(0, F.avg_pool2d(refs[0], 2, 2, count_include_pad=False))
        supps.insert(0, F.avg_pool2d(supps[0], 2, 2, count_include_pad=False))
    flow = refs[0].new_zeros([refs[0].size(0), 2, refs[0].size(2) // 2, refs[0].size(3) // 2])
    out = []
    for level in range(6):
        up = F.interpolate(flow, scale_factor=2, mode="bilinear", align_corners=True) * 2.0
        if up.size(2) != refs[level].size(2):                      # :119-122 (only the all-zero level-0 flow ever needs it)
            up = F.pad(up, [0, 0, 0, 1], mode="replicate")
        if up.size(3) != refs[level].size(3):
            up = F.pad(up, [0, 1, 0, 0], mode="replicate")
        x = _q(torch.cat([refs[level], flow_warp(supps[level], up, padding_mode="border"), up], 1))
        for j in (0, 2, 4, 6, 8):
            k = f"basic_module.{level}.basic_module.{j}."
            x = F.conv2d(x, _wq(sd[k + "weight"]), sd[k + "bias"], padding=3)
            if j < 8:
                x = _q(F.relu(x))
        flow = x + up
        if level in return_levels:
            scale = 2 ** (5 - level)
            fo = F.interpolate(flow, size=(h // scale, w // scale), mode="bilinear", align_corners=False)
            fo = fo * torch.tensor([float(w // scale) / float(w_floor // scale), float(h // scale) / float(h_floor // scale)],
                                   dtype=fo.dtype).view(1, 2, 1, 1)
            out.insert(0, fo)
    return out

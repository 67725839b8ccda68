"""VRT window attention on the MI355X (BASELINE config 5).

Drop-in for vsrlab ``src/vsr/models/VRT/modules/window_attention.py``: ``window_partition`` (:9-25),
``window_reverse`` (:27-43), ``get_window_size`` (:45-59), ``compute_mask`` (:61-77), ``Mlp_GEGLU`` (:79-98) and
``WindowAttention`` (:100-188) with the reference's constructor, parameter / buffer names and shapes
(``relative_position_bias_table``, ``relative_position_index``, ``qkv_self``, ``proj``, ``position_bias``, ``qkv_mut``).

What runs where: the attention itself -- (q*scale) k^T + relative-position bias + shift mask, softmax, attn @ v, for the
self attention and both mutual attentions, forward and backward -- is one fused MFMA kernel family
(csrc/window_attention.hip) that never writes the N x N scores; the qkv / proj ``nn.Linear`` layers are plain library
GEMMs (torch.nn.functional.linear = hipBLASLt), and the window partition / roll / mask construction is index glue."""
import math
from functools import lru_cache, reduce
from operator import mul

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..... import functional as VF


def window_partition(x, window_size):
    """(B, D, H, W, C) -> (B * windows, tokens per window, C): window-major, tokens in (d, h, w) order  (window_attention.py:9-24)."""
    B, C = x.shape[0], x.shape[-1]
    wd, wh, ww = window_size
    nd, nh, nw = x.shape[1] // wd, x.shape[2] // wh, x.shape[3] // ww
    tiles = x.reshape(B, nd, wd, nh, wh, nw, ww, C).permute(0, 1, 3, 5, 2, 4, 6, 7)     # (B, nd, nh, nw, wd, wh, ww, C)
    return tiles.reshape(B * nd * nh * nw, wd * wh * ww, C)


def window_reverse(windows, window_size, B, D, H, W):
    """inverse of window_partition: (B * windows, wd, wh, ww, C) -> (B, D, H, W, C)  (:26-42)."""
    wd, wh, ww = window_size
    nd, nh, nw = D // wd, H // wh, W // ww
    tiles = windows.reshape(B, nd, nh, nw, wd, wh, ww, -1).permute(0, 1, 4, 2, 5, 3, 6, 7)
    return tiles.reshape(B, D, H, W, -1)


def get_window_size(x_size, window_size, shift_size=None):
    """An axis not longer than its window is ONE window and is not shifted  (:44-59)."""
    fits = [extent <= win for extent, win in zip(x_size, window_size)]
    used = tuple(extent if f else win for extent, win, f in zip(x_size, window_size, fits))
    if shift_size is None:
        return used
    return used, tuple(0 if f else sh for sh, f in zip(shift_size, fits))


def _axis_regions(length, win, shift, device):
    """region label (0, 1, 2) of every position along one axis of the cyclically shifted volume: [0, L-win) | [L-win, L-shift) | [L-shift, L);
    with shift 0 the last region is the whole axis (what the reference's `slice(-0, None)` assignment leaves behind)"""
    pos = torch.arange(length, device=device)
    if shift == 0:
        return torch.full_like(pos, 2)
    return (pos >= length - win).long() + (pos >= length - shift).long()


@lru_cache()
def compute_mask(D, H, W, window_size, shift_size, device):
    """Attention mask of the shifted windows: 0 between tokens of the same region, -100 across regions  (:61-77)."""
    rd = _axis_regions(D, window_size[0], shift_size[0], device)
    rh = _axis_regions(H, window_size[1], shift_size[1], device)
    rw = _axis_regions(W, window_size[2], shift_size[2], device)
    region = ((rd[:, None, None] * 3 + rh[None, :, None]) * 3 + rw[None, None, :]).to(torch.float32).reshape(1, D, H, W, 1)
    per_window = window_partition(region, window_size).squeeze(-1)                        # (windows, tokens)
    differs = per_window[:, None, :] != per_window[:, :, None]
    return torch.where(differs, torch.full((), -100.0, device=device), torch.zeros((), device=device))


class Mlp_GEGLU(nn.Module):
    """GEGLU feed-forward (:79-98): fc2(gelu(fc11(x)) * fc12(x))."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc11 = nn.Linear(in_features, hidden_features)
        self.fc12 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.fc2(self.drop(self.act(self.fc11(x)) * self.fc12(x)))


class WindowAttention(nn.Module):
    """Window based multi-head mutual attention and self attention (:100-188)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=False, qk_scale=None, mut_attn=True):
        super().__init__()
        self.dim = dim
        self.window_size = window_size
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.mut_attn = mut_attn
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * window_size[0] - 1) * (2 * window_size[1] - 1) * (2 * window_size[2] - 1), num_heads))
        self.register_buffer("relative_position_index", self.get_position_index(window_size))
        self.qkv_self = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if self.mut_attn:
            self.register_buffer("position_bias", self.get_sine_position_encoding(window_size[1:], dim // 2, normalize=True))
            self.qkv_mut = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.proj = nn.Linear(2 * dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self.compute_dtype = None

    def forward(self, x, mask=None):
        """x: (num_windows*B, N, C); mask: (num_windows, N, N) of 0 / -100 or None."""
        if not x.is_cuda:
            raise RuntimeError("vsrlab_amd WindowAttention runs on MI355X only (no CPU fallback)")
        qkv_self = self.qkv_self(x)
        qkv_mut = self.qkv_mut(x + self.position_bias.repeat(1, 2, 1)) if self.mut_attn else None
        x_out = VF.window_attention_core(qkv_self, qkv_mut, self.relative_position_bias_table, self.relative_position_index, mask,
                                         self.num_heads, self.scale, self.compute_dtype)
        return self.proj(x_out)

    @staticmethod
    def get_position_index(window_size):
        """index into the bias table for every (query token, key token) pair of a window: the offset (dd, dh, dw) of the two tokens, each
        shifted to >= 0, in row-major order over (2 wd - 1, 2 wh - 1, 2 ww - 1)  (:164-179)"""
        wd, wh, ww = window_size
        d = torch.arange(wd).repeat_interleave(wh * ww)
        h = torch.arange(wh).repeat_interleave(ww).repeat(wd)
        w = torch.arange(ww).repeat(wd * wh)
        off = lambda v, n: v[:, None] - v[None, :] + (n - 1)            # noqa: E731
        return (off(d, wd) * (2 * wh - 1) + off(h, wh)) * (2 * ww - 1) + off(w, ww)

    @staticmethod
    def get_sine_position_encoding(HW, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        """2-D sine / cosine encoding of the (h, w) position of a window token, (1, H*W, 2 * num_pos_feats): y features first, even feature
        = sin, odd = cos, both of position / temperature^(2 (j // 2) / num_pos_feats)  (:181-188 and the helper it calls)"""
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        scale = 2 * math.pi if scale is None else scale
        H, W = HW
        ys = torch.arange(1, H + 1, dtype=torch.float32)
        xs = torch.arange(1, W + 1, dtype=torch.float32)
        if normalize:
            ys = ys / (ys[-1:] + 1e-6) * scale
            xs = xs / (xs[-1:] + 1e-6) * scale
        j = torch.arange(num_pos_feats, dtype=torch.float32)
        period = temperature ** (2 * (j // 2) / num_pos_feats)
        even = (torch.arange(num_pos_feats) % 2 == 0)

        def feats(p):                                                   # (L,) -> (L, num_pos_feats)
            a = p[:, None] / period
            return torch.where(even, a.sin(), a.cos())

        fy = feats(ys)[:, None, :].expand(H, W, num_pos_feats)
        fx = feats(xs)[None, :, :].expand(H, W, num_pos_feats)
        return torch.cat((fy, fx), dim=2).reshape(1, H * W, 2 * num_pos_feats).contiguous()

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
    """(B, D, H, W, C) -> (B*num_windows, Wd*Wh*Ww, C)."""
    B, D, H, W, C = x.shape
    x = x.view(B, D // window_size[0], window_size[0], H // window_size[1], window_size[1], W // window_size[2], window_size[2], C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, reduce(mul, window_size), C)


def window_reverse(windows, window_size, B, D, H, W):
    """(B*num_windows, Wd, Wh, Ww, C) -> (B, D, H, W, C)."""
    x = windows.view(B, D // window_size[0], H // window_size[1], W // window_size[2], window_size[0], window_size[1], window_size[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().view(B, D, H, W, -1)


def get_window_size(x_size, window_size, shift_size=None):
    use_window_size = list(window_size)
    use_shift_size = list(shift_size) if shift_size is not None else None
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            use_window_size[i] = x_size[i]
            if shift_size is not None:
                use_shift_size[i] = 0
    if shift_size is None:
        return tuple(use_window_size)
    return tuple(use_window_size), tuple(use_shift_size)


@lru_cache()
def compute_mask(D, H, W, window_size, shift_size, device):
    """Attention mask of the shifted windows: 0 inside a region, -100 across regions."""
    img_mask = torch.zeros((1, D, H, W, 1), device=device)
    cnt = 0
    for d in slice(-window_size[0]), slice(-window_size[0], -shift_size[0]), slice(-shift_size[0], None):
        for h in slice(-window_size[1]), slice(-window_size[1], -shift_size[1]), slice(-shift_size[1], None):
            for w in slice(-window_size[2]), slice(-window_size[2], -shift_size[2]), slice(-shift_size[2], None):
                img_mask[:, d, h, w, :] = cnt
                cnt += 1
    mask_windows = window_partition(img_mask, window_size).squeeze(-1)
    attn_mask = mask_windows.unsqueeze(1) - mask_windows.unsqueeze(2)
    return attn_mask.masked_fill(attn_mask != 0, float(-100.0)).masked_fill(attn_mask == 0, float(0.0))


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
        coords = torch.stack(torch.meshgrid(torch.arange(window_size[0]), torch.arange(window_size[1]), torch.arange(window_size[2]),
                                            indexing="ij"))
        coords_flatten = torch.flatten(coords, 1)
        relative_coords = (coords_flatten[:, :, None] - coords_flatten[:, None, :]).permute(1, 2, 0).contiguous()
        relative_coords[:, :, 0] += window_size[0] - 1
        relative_coords[:, :, 1] += window_size[1] - 1
        relative_coords[:, :, 2] += window_size[2] - 1
        relative_coords[:, :, 0] *= (2 * window_size[1] - 1) * (2 * window_size[2] - 1)
        relative_coords[:, :, 1] *= (2 * window_size[2] - 1)
        return relative_coords.sum(-1)

    @staticmethod
    def get_sine_position_encoding(HW, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        if scale is None:
            scale = 2 * math.pi
        not_mask = torch.ones([1, HW[0], HW[1]])
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if normalize:
            eps = 1e-6
            y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
            x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
        dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
        dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
        pos_embed = torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)
        return pos_embed.flatten(2).permute(0, 2, 1).contiguous()

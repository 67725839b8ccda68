"""``TMSA`` -- Temporal Mutual Self Attention block (vsrlab ``src/vsr/models/VRT/modules/tmsa.py:9-124``), with the
HIP window attention inside, and its two containers ``TMSAG`` (``tmsa.py:126-202``: depth blocks, every second one
shifted, one attention mask per call) and ``RTMSA`` (``tmsa.py:204-251``: residual TMSAG without mutual attention + a
linear layer).  Same constructors, parameter names (``norm1``, ``attn.*``, ``norm2``, ``mlp.*``; ``blocks.{i}.*``;
``residual_group.*``, ``linear.*``) and forwards.  ``drop_path`` > 0 = stochastic depth (``DropPath`` below, ``stochastic_depth.py``:
one per-sample Bernoulli scale of each residual branch in training, identity in eval)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .window_attention import Mlp_GEGLU, WindowAttention, compute_mask, get_window_size, window_partition, window_reverse


class DropPath(nn.Module):
    """Stochastic depth per sample (stochastic_depth.py:4-23): in training, a sample's residual branch is dropped with probability
    ``drop_prob`` and the kept ones are scaled by 1 / (1 - drop_prob); identity in eval.  One Bernoulli draw per sample from the CPU
    generator, exactly as the reference draws it (``torch.rand(shape)`` without a device), so a seeded run drops the same samples."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.drop_prob or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        draw = torch.rand((x.shape[0],) + (1,) * (x.ndim - 1))              # CPU generator, like the reference
        kept = torch.floor(keep + draw).to(device=x.device, dtype=x.dtype)      # 1 with probability keep, else 0
        return x.div(keep) * kept                                                 # the reference's operation order (bit-identical)


class TMSA(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=(6, 8, 8), shift_size=(0, 0, 0), mut_attn=True, mlp_ratio=2.,
                 qkv_bias=True, qk_scale=None, drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size = window_size, shift_size
        assert all(0 <= s < w for s, w in zip(shift_size, window_size)), "shift_size must be in range [0, window_size]"
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=self.window_size, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                    mut_attn=mut_attn)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp_GEGLU(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)

    def forward_part1(self, x, mask_matrix):
        """norm1 -> pad the volume to whole windows -> (cyclic shift) -> windows -> attention -> back (tmsa.py:60-105)"""
        B, D, H, W, C = x.shape
        win, shift = get_window_size((D, H, W), self.window_size, self.shift_size)
        y = self.norm1(x)
        pad = [(-extent) % w for extent, w in zip((D, H, W), win)]              # trailing padding per axis
        if any(pad):
            y = F.pad(y, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
        Dp, Hp, Wp = D + pad[0], H + pad[1], W + pad[2]
        shifted = any(sh > 0 for sh in shift)
        if shifted:
            y = torch.roll(y, shifts=tuple(-sh for sh in shift), dims=(1, 2, 3))
        out = self.attn(window_partition(y, win), mask=mask_matrix if shifted else None)
        y = window_reverse(out.view(-1, *win, C), win, B, Dp, Hp, Wp)
        if shifted:
            y = torch.roll(y, shifts=tuple(shift), dims=(1, 2, 3))
        return self.drop_path(y[:, :D, :H, :W, :] if any(pad) else y)

    def forward_part2(self, x):
        return self.drop_path(self.mlp(self.norm2(x)))

    def forward(self, x, mask_matrix):
        x = x + self.forward_part1(x, mask_matrix)
        return x + self.forward_part2(x)


class TMSAG(nn.Module):
    """Temporal Mutual Self Attention Group (tmsa.py:126-202).  forward(x): (B, C, D, H, W) -> (B, C, D, H, W)."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size=[6, 8, 8], shift_size=None, mut_attn=True, mlp_ratio=2.,
                 qkv_bias=False, qk_scale=None, drop_path=0., norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution = input_resolution
        self.window_size = window_size
        self.shift_size = list(i // 2 for i in window_size) if shift_size is None else shift_size
        self.blocks = nn.ModuleList([
            TMSA(dim=dim, input_resolution=input_resolution, num_heads=num_heads, window_size=tuple(window_size),
                 shift_size=(0, 0, 0) if i % 2 == 0 else tuple(self.shift_size), mut_attn=mut_attn, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                 qk_scale=qk_scale, drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer)
            for i in range(depth)])

    def forward(self, x):
        B, C, D, H, W = x.shape
        window_size, shift_size = get_window_size((D, H, W), tuple(self.window_size), tuple(self.shift_size))
        x = x.permute(0, 2, 3, 4, 1)
        Dp = -(-D // window_size[0]) * window_size[0]
        Hp = -(-H // window_size[1]) * window_size[1]
        Wp = -(-W // window_size[2]) * window_size[2]
        # one mask for all shifted blocks (:191).  The reference casts it `.type_as(x)`; the HIP attention reads masks as fp32
        # (0 / -100 are exact in every dtype), and passing the lru-cached tensor itself lets functional._packed_mask hit its
        # cache on every block of every forward (a fresh cast per call defeated it: two host syncs per attention call)
        attn_mask = compute_mask(Dp, Hp, Wp, window_size, shift_size, x.device)
        for blk in self.blocks:
            x = blk(x, attn_mask)
        return x.reshape(B, D, H, W, -1).permute(0, 4, 1, 2, 3)


class RTMSA(nn.Module):
    """Residual Temporal Mutual Self Attention (tmsa.py:204-251): x + linear(TMSAG(x)) with self attention only."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=2., qkv_bias=True, qk_scale=None, drop_path=0.,
                 norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.residual_group = TMSAG(dim=dim, input_resolution=input_resolution, depth=depth, num_heads=num_heads, window_size=window_size,
                                    mut_attn=False, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop_path=drop_path,
                                    norm_layer=norm_layer)
        self.linear = nn.Linear(dim, dim)

    def forward(self, x):
        return x + self.linear(self.residual_group(x).transpose(1, 4)).transpose(1, 4)

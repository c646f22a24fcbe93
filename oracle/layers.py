"""Oracle layer library: fp32 CPU restatement of the diffusers==0.24.0 modules
that the reference composes (imports at /root/reference/models/unet_3d_blocks.py:18-20
and /root/reference/models/unet_3d_condition_mask.py:22-26).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Attribute names reproduce the
diffusers state-dict key layout so one seeded state dict
loads into both this oracle and the HIP product modules.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- embeddings
def sinusoid_embedding(t: torch.Tensor, dim: int = 320, flip_sin_to_cos: bool = True,
                       downscale_freq_shift: float = 0.0, max_period: float = 10000.0) -> torch.Tensor:
    """diffusers `Timesteps` / `get_timestep_embedding`; used at
    unet_3d_condition_mask.py:146,156 (time_proj / motion_proj)."""
    half = dim // 2
    expo = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device)
    freq = torch.exp(expo / (half - downscale_freq_shift))
    arg = t.reshape(-1).float()[:, None] * freq[None, :]
    emb = torch.cat([arg.sin(), arg.cos()], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


class TimestepEmbedding(nn.Module):
    """diffusers `TimestepEmbedding(in, dim, act='silu', cond_proj_dim=...)`;
    constructed at unet_3d_condition_mask.py:149-154."""

    def __init__(self, in_channels: int, time_embed_dim: int, cond_proj_dim: int | None = None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.cond_proj = nn.Linear(cond_proj_dim, in_channels, bias=False) if cond_proj_dim else None
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        if condition is not None:
            sample = sample + self.cond_proj(condition)
        return self.linear_2(F.silu(self.linear_1(sample)))


# --------------------------------------------------------------------------- conv blocks
class ResnetBlock2D(nn.Module):
    """diffusers `ResnetBlock2D`, time_embedding_norm='default', pre_norm;
    constructed at unet_3d_blocks.py:262,309,425,564,660,794."""

    def __init__(self, in_channels, out_channels=None, temb_channels=1280, eps=1e-5, groups=32,
                 output_scale_factor=1.0):
        super().__init__()
        out_channels = out_channels or in_channels
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.output_scale_factor = output_scale_factor

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None and temb is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        skip = x if self.conv_shortcut is None else self.conv_shortcut(x)
        return (skip + h) / self.output_scale_factor


class TemporalConvLayer(nn.Module):
    """diffusers `TemporalConvLayer`: 4 x {GroupNorm(32) over (C/32,T,H,W), SiLU,
    Conv3d (3,1,1)} + identity; the last conv is zero-initialised.  Constructed at
    unet_3d_blocks.py:276,323,439,578,674,808."""

    def __init__(self, in_dim, out_dim=None, dropout=0.0):
        super().__init__()
        out_dim = out_dim or in_dim

        def conv():
            return nn.Conv3d(out_dim, in_dim, (3, 1, 1), padding=(1, 0, 0))

        self.conv1 = nn.Sequential(nn.GroupNorm(32, in_dim), nn.SiLU(),
                                   nn.Conv3d(in_dim, out_dim, (3, 1, 1), padding=(1, 0, 0)))
        self.conv2 = nn.Sequential(nn.GroupNorm(32, out_dim), nn.SiLU(), nn.Dropout(dropout), conv())
        self.conv3 = nn.Sequential(nn.GroupNorm(32, out_dim), nn.SiLU(), nn.Dropout(dropout), conv())
        self.conv4 = nn.Sequential(nn.GroupNorm(32, out_dim), nn.SiLU(), nn.Dropout(dropout), conv())
        nn.init.zeros_(self.conv4[-1].weight)
        nn.init.zeros_(self.conv4[-1].bias)

    def forward(self, x, num_frames=1):
        bt, c, h, w = x.shape
        x5 = x.reshape(bt // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        y = self.conv4(self.conv3(self.conv2(self.conv1(x5))))
        y = x5 + y
        return y.permute(0, 2, 1, 3, 4).reshape(bt, c, h, w)


class Downsample2D(nn.Module):
    """diffusers `Downsample2D(use_conv=True)`.  padding=0 is the VAE-encoder
    form: F.pad (0,1,0,1) then a stride-2 conv."""

    def __init__(self, channels, out_channels=None, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1))
        return self.conv(x)


class Upsample2D(nn.Module):
    """diffusers `Upsample2D(use_conv=True)`: nearest x2 (or to `output_size`)
    followed by a 3x3 conv."""

    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=tuple(output_size), mode="nearest")
        return self.conv(x)


# --------------------------------------------------------------------------- attention
class Attention(nn.Module):
    """diffusers `Attention` with `AttnProcessor2_0` semantics (reference
    train.py:124-138): softmax(q k^T / sqrt(d)) v per head, no mask, no dropout.
    `group_norm`/`residual_connection`/`bias` cover the VAE mid-block form (A.9)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False,
                 norm_num_groups=None, eps=1e-5, residual_connection=False):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.residual_connection = residual_connection
        self.group_norm = nn.GroupNorm(norm_num_groups, query_dim, eps=eps) if norm_num_groups else None
        kv_dim = cross_attention_dim or query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])

    def forward(self, hidden_states, encoder_hidden_states=None):
        x = hidden_states
        spatial = x.dim() == 4
        if spatial:
            b, c, hh, ww = x.shape
            x = x.reshape(b, c, hh * ww).transpose(1, 2)
        residual = x
        if self.group_norm is not None:
            x = self.group_norm(x.transpose(1, 2)).transpose(1, 2)
        ctx = x if encoder_hidden_states is None else encoder_hidden_states
        b, lq, _ = x.shape
        q = self.to_q(x).reshape(b, lq, self.heads, -1).transpose(1, 2)
        k = self.to_k(ctx).reshape(b, ctx.shape[1], self.heads, -1).transpose(1, 2)
        v = self.to_v(ctx).reshape(b, ctx.shape[1], self.heads, -1).transpose(1, 2)
        scores = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
        out = torch.matmul(scores.softmax(dim=-1), v)
        out = out.transpose(1, 2).reshape(b, lq, -1)
        out = self.to_out[0](out)
        if self.residual_connection:
            out = out + residual
        if spatial:
            out = out.transpose(1, 2).reshape(b, c, hh, ww)
        return out


class GEGLU(nn.Module):
    """diffusers `GEGLU`: proj to 2*d, value * gelu_erf(gate)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        val, gate = self.proj(x).chunk(2, dim=-1)
        return val * F.gelu(gate)


class FeedForward(nn.Module):
    """diffusers `FeedForward(dim, mult=4, activation_fn='geglu')`."""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    """diffusers `BasicTransformerBlock`: LN->attn1, LN->attn2, LN->GEGLU FF, each
    with a residual.  `double_self_attention` makes attn2 a second self-attention."""

    def __init__(self, dim, heads, head_dim, cross_attention_dim=None, double_self_attention=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, head_dim)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, None if double_self_attention else cross_attention_dim, heads, head_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)
        self.double_self_attention = double_self_attention

    def forward(self, x, encoder_hidden_states=None):
        x = x + self.attn1(self.norm1(x))
        ctx = None if self.double_self_attention else encoder_hidden_states
        x = x + self.attn2(self.norm2(x), ctx)
        x = x + self.ff(self.norm3(x))
        return x


class _Sample:
    def __init__(self, sample):
        self.sample = sample


class Transformer2DModel(nn.Module):
    """diffusers `Transformer2DModel(use_linear_projection=True)`; the reference's
    block factories force linear projections (unet_3d_blocks.py:136,192,250) and call it at
    unet_3d_blocks.py:372-376,519-523,752-756."""

    def __init__(self, heads, head_dim, in_channels, cross_attention_dim=1024, norm_num_groups=32):
        super().__init__()
        inner = heads * head_dim
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, head_dim, cross_attention_dim)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, encoder_hidden_states=None):
        n, c, h, w = x.shape
        res = x
        t = self.norm(x).permute(0, 2, 3, 1).reshape(n, h * w, c)
        t = self.proj_in(t)
        for blk in self.transformer_blocks:
            t = blk(t, encoder_hidden_states)
        t = self.proj_out(t)
        t = t.reshape(n, h, w, c).permute(0, 3, 1, 2)
        return _Sample(t + res)


class TransformerTemporalModel(nn.Module):
    """diffusers `TransformerTemporalModel(double_self_attention=True)`: 5-D
    GroupNorm over (C/32,T,H,W), tokens = frames, batch = B*H*W; called at
    unet_3d_condition_mask.py:437 and unet_3d_blocks.py:379,526,759 without text."""

    def __init__(self, heads, head_dim, in_channels, norm_num_groups=32):
        super().__init__()
        inner = heads * head_dim
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, head_dim, None, double_self_attention=True)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, num_frames=1):
        bt, c, h, w = x.shape
        b = bt // num_frames
        res = x
        x5 = x.reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        x5 = self.norm(x5)
        t = x5.permute(0, 3, 4, 2, 1).reshape(b * h * w, num_frames, c)
        t = self.proj_in(t)
        for blk in self.transformer_blocks:
            t = blk(t)
        t = self.proj_out(t)
        t = t.reshape(b, h, w, num_frames, c).permute(0, 3, 4, 1, 2).reshape(bt, c, h, w)
        return _Sample(t + res)

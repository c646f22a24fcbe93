"""HIP-backed layer library: the diffusers==0.24.0 building blocks the reference composes
(/root/reference/models/unet_3d_blocks.py:18-20, /root/reference/models/unet_3d_condition_mask.py:22-26),
re-designed around channels-last token matrices.

Every module keeps the parameter names of its diffusers counterpart (state-dict compatible, SURVEY.md
Appendix C) but computes through `ops` (libaa_mi355.so).  Activations travel between modules as
`[tokens, C]` matrices plus a `Grid` describing (clips B, frames T, H, W); nothing is ever permuted:
the spatial view (image n = b*T+t, pixel) and the temporal view (clip b, pixel, frame) are two ways
of indexing the same buffer.
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Optional

import os

import torch
import torch.nn as nn

from . import ops
from ._lib import AA_ACT_NONE, AA_ACT_SILU


@dataclass(frozen=True)
class Grid:
    """Geometry of a token matrix: tokens = clips*frames*h*w, ordered (clip, frame, y, x)."""
    clips: int
    frames: int
    h: int
    w: int

    @property
    def images(self):
        return self.clips * self.frames

    @property
    def hw(self):
        return self.h * self.w

    @property
    def tokens(self):
        return self.images * self.hw

    def resized(self, h, w):
        return replace(self, h=h, w=w)


def weights_key(*tensors):
    """Identity + in-place version of the source parameters of a packed-weight cache: a cache built from other storage, or
    from the same storage before an in-place update (load_state_dict copies in place, LoRA merges, optimiser steps), is stale."""
    return tuple((t.data_ptr(), t._version) for t in tensors if t is not None)


class _Packed:
    """Mixin: caches the kernel-side weight layout, keyed on the parameters' storage and in-place version."""
    _pw = None
    _pw_key = None

    def _apply(self, fn, *a, **k):
        self._pw = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._pw = None
        return super()._load_from_state_dict(*a, **k)

    def packed(self):
        key = weights_key(self.weight, self.bias)
        if self._pw is None or self._pw_key != key:
            self._pw, self._pw_key = ops.pack_weight(self.weight, self.bias), key
        return self._pw


LN_FOLD = True       # BasicTransformerBlock: LayerNorm folded into the contraction that consumes it (ops.pack_weight(ln=...)), the row
                     # statistics taken from the epilogue of the contraction that produced the tensor; False = LayerNorm kernels
# ... norm3 -> GEGLU as well: the fold costs the GEGLU projection 10 % at the 64x64 level (two accumulator blocks per output block to
# start and to scale: 3.29 -> 3.61 ms per step there, +0.07 / +0.03 ms at the next levels) and saves the LayerNorm kernels in front
# (0.36 / 0.10 / 0.03 ms) plus 1/3 of the finalize launches: an A/B on one box gave 64.57 / 64.63 ms with, 64.80 / 64.72 ms without
# (profiles/r04q_*).  On (equal time, 3.4 GB / step less traffic); AA_LN_FOLD_FF=0 switches it off.
LN_FOLD_FF = os.environ.get("AA_LN_FOLD_FF", "1") == "1"
# Transformer2DModel / TransformerTemporalModel end with  proj_out(ff_out(h) + x) + residual  (h = the GEGLU output, x = the block's
# residual stream): two linear maps with only an addition between them.  Merged into ONE contraction over the channel-concatenated
# sources [h | x] with the weights [Wp W2 | Wp] and the bias Wp b2 + bp (products in fp32, one rounding to the storage type - the
# same kind of exact re-association as the LayerNorm fold): the same 2 M (4C + C) C multiply-adds, but the K = C projection - bound by
# its three tensor passes and per-tile fixed costs, 420-730 TF/s - rides as 25 % more K on a K = 4C contraction that runs at 730-950,
# and the block output is never written or re-read (-2 tensor passes, -30 launches per step).  AA_FF_PROJ_MERGE=0: the two calls.
FF_PROJ_MERGE = os.environ.get("AA_FF_PROJ_MERGE", "1") == "1"
# The self-attention layers of TransformerTemporalModel (17-frame sequences): LayerNorm, the Q|K|V projection and the attention as ONE
# kernel (ops.seq_self_attention, ABI 108) - Q|K|V are never written.  AA_SEQ_ATTN=0: LayerNorm fold + Q|K|V contraction + aa_attention.
SEQ_ATTN = os.environ.get("AA_SEQ_ATTN", "1") == "1"
# ... where it is faster than the three launches: 320 channels (two workgroups per CU: 135-160 us against 262-278 at the 64x64 level) and 512
# (transformer_in: 347-384 against 491-518); at 640 channels (32x32 level) x takes 160 registers, one workgroup per CU is left and 293 tiles
# make two rounds of 256: 194-211 us against 163-174 (profiles/r06b_seq_attention_two_per_cu.txt) - the three launches stay there
# ... and the linear layer in front of such an attention layer inside the same kernel (ops.seq_self_attention(pre=...)): proj_in in front of
# the first one, the first one's to_out + residual in front of the second (AA_SEQ_PRE=0: separate contractions)
SEQ_PRE = os.environ.get("AA_SEQ_PRE", "0") == "1"   # measured (r06f): 214 us against 151 + 79 apart in isolation, but +0.1 ms on the step - off
# FeedForward + proj_out of a transformer as ONE kernel where the library has it (320 channels: the 64x64 level): ops.ff_fused instead of the GEGLU
# contraction + the merged ff-out / proj_out contraction - the [tokens, 4 C] activation is never written (AA_FF_FUSED=0: the two contractions).
# AA_FF_SPLIT=1: rows beyond the last FULL round of 128-row tiles over the chip's CUs (1088 tiles on 256 CUs = 4.25 rounds at the 64x64 level:
# the fifth round runs 25 % full) go through the two contractions instead.
FF_FUSED = os.environ.get("AA_FF_FUSED", "1") == "1"
FF_SPLIT = os.environ.get("AA_FF_SPLIT", "1") == "1"
SEQ_ATTN_CHANNELS = tuple(int(c) for c in os.environ.get("AA_SEQ_ATTN_CHANNELS", "320,512").split(",") if c)
# The K = C projections of the 320-channel transformers (proj_in, to_q, to_out + residual, the Q|K|V of the spatial self-attention) with the token
# rows in registers (ops.linear_rows, ABI 109) instead of tiles of the contraction family: the rows are normalised in the registers, so the
# producers in front emit no row statistics.  AA_LINEAR_ROWS=0: the contractions (LayerNorm folded from producer-written coefficients).
LINEAR_ROWS = os.environ.get("AA_LINEAR_ROWS", "1") == "1"
# ... and the GroupNorm in front of proj_in (affine-only: no SiLU behind it) applied to the rows in those registers: its statistics pass runs alone
# (ops.groupnorm_coef), the normalised tensor is never written (AA_GN_FOLD=0: the statistics + normalise pair in front of proj_in).
GN_FOLD = os.environ.get("AA_GN_FOLD", "1") == "1"
LINEAR_ROWS_MIN = 16384              # below: a handful of 128-row workgroups that walk all output channels serially - the tile family is faster


def rows_path(channels, n_out, rows, dtype):
    return LINEAR_ROWS and rows >= LINEAR_ROWS_MIN and ops.linear_rows_ok(channels, n_out, rows, dtype)


UPSAMPLE_AS_PARITY_CONVS = True      # Upsample2D at exactly x2: four 2x2 convolutions (ops.pack_upsample2x_weights); False = the 3x3 gather form


def _no_eager(name):
    raise RuntimeError(f"{name}: this module only runs through the HIP token path of animate_anything_amd")


class Linear(_Packed, nn.Linear):
    _sp = None
    _sp_key = None

    _rp = None
    _rp_key = None

    def rows_packed(self):
        """This layer as the weight stream of ops.linear_rows, cached."""
        key = weights_key(self.weight, self.bias)
        if self._rp is None or self._rp_key != key:
            self._rp, self._rp_key = ops.pack_linear_rows(self.weight, self.bias), key
        return self._rp

    def tokens(self, x, affine=None, **epilogue):
        """`affine` = (ops.groupnorm_coef(...), rows per image group) of the GroupNorm in front (caller checked norm_rows_ok): x is the un-normalised tensor."""
        if affine is not None or (not epilogue.get("row_stats") and not (set(epilogue) - {"residual", "row_stats", "coef_eps"}) and x.dim() == 2
                                  and x.stride(1) == 1 and rows_path(self.in_features, self.out_features, x.shape[0], x.dtype) and x.shape[1] == self.in_features):
            return ops.linear_rows(x, self.rows_packed(), epilogue.get("residual"), affine=affine)
        return ops.conv_gemm(x, self.packed(), ops.linear_geom(x.shape[0]), **epilogue)

    def norm_rows_ok(self, x, tokens_per_group):
        """Can a GroupNorm in front of this layer be applied inside it (ops.linear_rows(affine=...): its normalisation pass is never run)?"""
        return GN_FOLD and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == self.in_features and \
            rows_path(self.in_features, self.out_features, x.shape[0], x.dtype) and ops.linear_rows_ok(self.in_features, self.out_features, x.shape[0], x.dtype, tokens_per_group)

    def seq_pre(self):
        """This layer as the projection run inside ops.seq_self_attention in front of the normalisation (ops.pack_seq_pre), cached."""
        key = weights_key(self.weight, self.bias)
        if self._sp is None or self._sp_key != key:
            self._sp, self._sp_key = ops.pack_seq_pre(self.weight, self.bias), key
        return self._sp

    def _apply(self, fn, *a, **k):
        self._sp = None
        self._rp = None
        return super()._apply(fn, *a, **k)

    def forward(self, x):
        _no_eager("Linear")


class Conv2d(_Packed, nn.Conv2d):
    def tokens(self, x, geom, **epilogue):
        return ops.conv_gemm(x, self.packed(), geom, **epilogue)

    def forward(self, x):
        _no_eager("Conv2d")


class Conv3d(_Packed, nn.Conv3d):
    def tokens(self, x, geom, **epilogue):
        return ops.conv_gemm(x, self.packed(), geom, **epilogue)

    def forward(self, x):
        _no_eager("Conv3d")


class GroupNorm(nn.GroupNorm):
    def coef_tokens(self, x, n_stat_groups, tokens_per_group):
        """The statistics alone, as per-channel (scale, shift) of every image group (ops.groupnorm_coef): the consumer normalises."""
        return ops.groupnorm_coef(x, self.weight, self.bias, n_stat_groups, tokens_per_group, self.num_groups, self.eps)

    def tokens(self, x, n_stat_groups, tokens_per_group, silu=False, x1=None, out=None):
        return ops.groupnorm(x, self.weight, self.bias, n_stat_groups, tokens_per_group, self.num_groups,
                             self.eps, silu, x1=x1, out=out)

    def forward(self, x):
        _no_eager("GroupNorm")


class LayerNorm(nn.LayerNorm):
    def tokens(self, x):
        return ops.layernorm(x, self.weight, self.bias, self.eps)

    def forward(self, x):
        _no_eager("LayerNorm")


# ------------------------------------------------------------------------------------- embeddings
class TimestepEmbedding(nn.Module):
    """diffusers TimestepEmbedding(act='silu', cond_proj_dim, out_dim).  `tokens` returns
    silu(linear_2(...)) when `final_silu` since the only consumers are the resnets'
    time_emb_proj(silu(temb))."""

    def __init__(self, in_channels, time_embed_dim, cond_proj_dim=None, out_dim=None):
        super().__init__()
        self.linear_1 = Linear(in_channels, time_embed_dim)
        self.cond_proj = Linear(cond_proj_dim, in_channels, bias=False) if cond_proj_dim else None
        self.linear_2 = Linear(time_embed_dim, out_dim or time_embed_dim)

    def tokens(self, sample, condition=None, final_silu=False):
        if condition is not None:
            sample = self.cond_proj.tokens(condition, residual=sample)
        h = self.linear_1.tokens(sample, act=AA_ACT_SILU)
        return self.linear_2.tokens(h, act=AA_ACT_SILU if final_silu else AA_ACT_NONE)


# ------------------------------------------------------------------------------------- conv blocks
class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D.  The input may arrive as two channel-concatenated
    sources (up-block skip connections) - the concat is never materialised."""

    def __init__(self, in_channels, out_channels=None, temb_channels=1280, eps=1e-5, groups=32,
                 output_scale_factor=1.0):
        super().__init__()
        out_channels = out_channels or in_channels
        self.norm1 = GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = Linear(temb_channels, out_channels) if temb_channels else None
        self.norm2 = GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.output_scale_factor = output_scale_factor
        self.tproj = None        # this block's slice of a batched time-embedding projection (set by the UNet per forward)

    def tokens(self, x, g: Grid, temb_silu=None, x1=None):
        geom = ops.conv3x3_geom(g.images, g.h, g.w)
        h = self.norm1.tokens(x, g.images, g.hw, silu=True, x1=x1)
        if self.time_emb_proj is not None and temb_silu is not None:
            # [clips, Cout]: normally a column slice of ONE projection of all the blocks' time embeddings
            tproj = self.tproj if self.tproj is not None else self.time_emb_proj.tokens(temb_silu)
            self.tproj = None
            h = self.conv1.tokens(h, geom, rowvec=tproj, rowvec_div=g.frames * g.hw)
        else:
            h = self.conv1.tokens(h, geom)
        h = self.norm2.tokens(h, g.images, g.hw, silu=True)
        if self.conv_shortcut is not None:
            skip = self.conv_shortcut.tokens(x, ops.linear_geom(g.tokens), x1=x1)
        else:
            skip = x
        return self.conv2.tokens(h, geom, residual=skip, out_scale=1.0 / self.output_scale_factor)


class TemporalConvLayer(nn.Module):
    """diffusers TemporalConvLayer: 4 x {GroupNorm over (C/32,T,H,W), SiLU, Conv3d (3,1,1)}
    + identity.  The clip-wide statistics are one GroupNorm over T*H*W consecutive tokens; the
    temporal conv is a 3x1 implicit GEMM over the (frame, pixel) grid - no [B,C,T,H,W] permutes."""

    def __init__(self, in_dim, out_dim=None, dropout=0.0):
        super().__init__()
        out_dim = out_dim or in_dim

        def stage(cin, cout, with_dropout):
            mods = [GroupNorm(32, cin), nn.SiLU()] + ([nn.Dropout(dropout)] if with_dropout else [])
            return nn.Sequential(*mods, Conv3d(cin, cout, (3, 1, 1), padding=(1, 0, 0)))

        self.conv1 = stage(in_dim, out_dim, False)
        self.conv2 = stage(out_dim, in_dim, True)
        self.conv3 = stage(out_dim, in_dim, True)
        self.conv4 = stage(out_dim, in_dim, True)
        nn.init.zeros_(self.conv4[-1].weight)
        nn.init.zeros_(self.conv4[-1].bias)

    def tokens(self, x, g: Grid):
        geom = ops.tconv_geom(g.clips, g.frames, g.hw)
        h = x
        for i, seq in enumerate((self.conv1, self.conv2, self.conv3, self.conv4)):
            h = seq[0].tokens(h, g.clips, g.frames * g.hw, silu=True)
            h = seq[-1].tokens(h, geom, residual=x if i == 3 else None)
        return h


class Downsample2D(nn.Module):
    def __init__(self, channels, out_channels=None, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def tokens(self, x, g: Grid):
        geom = ops.conv3x3_geom(g.images, g.h, g.w, stride=2, pad=self.padding)
        return self.conv.tokens(x, geom), g.resized(geom.h_out, geom.w_out)


class Upsample2D(nn.Module):
    """Nearest x2 (or to `output_size`) fused into the 3x3 conv's gather."""

    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.conv = Conv2d(channels, out_channels or channels, 3, padding=1)

    def parity_packed(self):
        """The four 2x2 filters of the exact x2 form (ops.pack_upsample2x_weights), cached like Conv2d.packed()."""
        key = weights_key(self.conv.weight, self.conv.bias)
        if self._pp is None or self._pp_key != key:
            self._pp, self._pp_key = ops.pack_upsample2x_weights(self.conv.weight, self.conv.bias), key
        return self._pp

    _pp = None
    _pp_key = None

    def _apply(self, fn, *a, **k):
        self._pp = None
        return super()._apply(fn, *a, **k)

    def tokens(self, x, g: Grid, output_size=None):
        up = (2 * g.h, 2 * g.w) if output_size is None else tuple(int(s) for s in output_size)
        c = x.shape[-1]
        if up == (2 * g.h, 2 * g.w) and c % 64 == 0 and self.conv.out_channels % 8 == 0 and UPSAMPLE_AS_PARITY_CONVS:
            # exact x2: four 2x2 convolutions on the stored grid, each writing one output parity class (4/9 of the multiply-adds)
            out = torch.empty(g.images * up[0] * up[1], self.conv.out_channels, dtype=x.dtype, device=x.device)
            for (a, b), pw in self.parity_packed().items():
                geom = ops.Geom(g.images, g.h, g.w, g.h, g.w, 1, 1 - a, 1 - b)
                ops.conv_gemm(x, pw, geom, out=out, out_map=(2, 2, a, b))
            return out, g.resized(*up)
        geom = ops.conv3x3_geom(g.images, g.h, g.w, up_to=up)
        return self.conv.tokens(x, geom), g.resized(*up)


# ------------------------------------------------------------------------------------- attention
class Attention(nn.Module):
    """diffusers Attention + AttnProcessor2_0, head_dim 64.  Q/K/V of a self-attention
    come from ONE GEMM against the row-concatenated weights; the text K/V of a cross-attention are
    computed once per clip (the reference repeats the text per frame, unet_3d_condition_mask.py:421)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.inner = heads, dim_head, inner
        self.is_cross = cross_attention_dim is not None
        kv_dim = cross_attention_dim or query_dim
        self.to_q = Linear(query_dim, inner, bias=False)
        self.to_k = Linear(kv_dim, inner, bias=False)
        self.to_v = Linear(kv_dim, inner, bias=False)
        self.to_out = nn.ModuleList([Linear(inner, query_dim), nn.Dropout(0.0)])
        self._fused = None
        self._fused_key = None
        self.kv = None

    def _apply(self, fn, *a, **k):
        self._fused = None
        self._fused_ln = None
        self._rows_ln = None
        self._seq_w = None
        return super()._apply(fn, *a, **k)

    def fused(self):
        rows = [self.to_k.weight, self.to_v.weight] if self.is_cross else \
               [self.to_q.weight, self.to_k.weight, self.to_v.weight]
        key = weights_key(*rows)
        if self._fused is None or self._fused_key != key:
            self._fused, self._fused_key = ops.pack_weight(torch.cat([r.detach() for r in rows], dim=0)), key
        return self._fused

    def fused_ln(self, norm):
        """The Q|K|V rows (self-attention) or the Q rows (cross-attention) with `norm` - the LayerNorm in front - folded in."""
        rows = [self.to_q.weight] if self.is_cross else [self.to_q.weight, self.to_k.weight, self.to_v.weight]
        key = weights_key(*rows, norm.weight, norm.bias)
        if self._fused_ln is None or self._fused_ln_key != key:
            self._fused_ln = ops.pack_weight(torch.cat([r.detach() for r in rows], dim=0), None, ln=(norm.weight, norm.bias, norm.eps))
            self._fused_ln_key = key
        return self._fused_ln

    def rows_ln(self, norm):
        """fused_ln(norm) as the weight stream of ops.linear_rows (the rows are normalised inside the kernel)."""
        rows = [self.to_q.weight] if self.is_cross else [self.to_q.weight, self.to_k.weight, self.to_v.weight]
        key = weights_key(*rows, norm.weight, norm.bias)
        if self._rows_ln is None or self._rows_ln_key != key:
            self._rows_ln = ops.pack_linear_rows(torch.cat([r.detach() for r in rows], dim=0), None, ln=(norm.weight, norm.bias, norm.eps))
            self._rows_ln_key = key
        return self._rows_ln

    def rows_ok(self, x):
        """Does the projection behind the LayerNorm in front of this layer (Q|K|V, or Q of a cross-attention) run on ops.linear_rows?"""
        return self.to_q.bias is None and x.dim() == 2 and x.stride(1) == 1 and \
            rows_path(self.to_q.in_features, self.inner * (1 if self.is_cross else 3), x.shape[0], x.dtype)

    _rows_ln = None
    _rows_ln_key = None
    _fused_ln = None
    _fused_ln_key = None
    _seq_w = None
    _seq_w_key = None

    def seq_packed(self, norm):
        """The operands of ops.seq_self_attention: per head the to_k, to_v (permuted) and to_q rows with `norm` - the LayerNorm in front - folded in."""
        key = weights_key(self.to_q.weight, self.to_k.weight, self.to_v.weight, norm.weight, norm.bias)
        if self._seq_w is None or self._seq_w_key != key:
            self._seq_w = ops.pack_seq_qkv(self.to_q.weight, self.to_k.weight, self.to_v.weight, ln=(norm.weight, norm.bias, norm.eps))
            self._seq_w_key = key
        return self._seq_w

    def seq_ok(self, x, g: "Grid", temporal: bool):
        """Can this self-attention over the frames of a pixel run as ops.seq_self_attention (LayerNorm + Q|K|V + attention in one kernel)?"""
        return (SEQ_ATTN and temporal and not self.is_cross and self.dim_head == 64 and self.inner == x.shape[1] == self.to_q.in_features
                and self.inner in SEQ_ATTN_CHANNELS and self.to_q.bias is None and ops.seq_self_attention_ok(self.inner, g.frames, x.shape[0], x.dtype))

    def seq_attention(self, x, norm, g: "Grid", pre=None, residual=None):
        """softmax(q k^T) v over the frames of each pixel with [q | k | v] = norm(x') W^T, x' = x (or pre(x) + residual) - one kernel, the
        output projection NOT included.  Returns o, or (o, x') with `pre` (a Linear)."""
        return ops.seq_self_attention(x, self.seq_packed(norm), g.clips, g.hw, g.frames, (g.frames * g.hw, 1, g.hw), scale=float(self.dim_head) ** -0.5,
                                      pre=None if pre is None else pre.seq_pre(), residual=residual)

    def text_kv(self, text_tokens):
        """[clips*L, cross_dim] -> [clips*L, 2*inner] (K | V): normally a column slice of ONE projection of the text for
        all cross-attention layers of the network (`kv` set by the UNet per forward)."""
        if self.kv is not None:
            kv, self.kv = self.kv, None
            return kv
        return ops.conv_gemm(text_tokens, self.fused(), ops.linear_geom(text_tokens.shape[0]))

    def cross_rowvec(self, text_tokens):
        """A ONE-token context (the CLIP image embedding of the SVD path): softmax over a single key is exactly 1, the
        cross-attention output of every query of clip b is to_out(to_v(ctx_b)) - a row vector per clip, [clips, query_dim]."""
        kv = self.text_kv(text_tokens)
        return self.to_out[0].tokens(kv[:, self.inner:].contiguous())

    def self_tokens(self, normed, residual, g: Grid, temporal: bool, ln=None, seq_ln=None, rows_ln=None, **epilogue):
        """`ln` = (LayerNorm module, ops.RowStats of `normed`'s rows): `normed` is then the UN-normalised tensor and the
        LayerNorm is folded into the Q|K|V projection.  `row_stats=True` (epilogue) returns (out, RowStats or None).
        `seq_ln` = the LayerNorm module in front (caller checked seq_ok): `normed` is the un-normalised tensor; LayerNorm, Q|K|V
        and the attention over the frames of each pixel run as one kernel, only the output projection follows."""
        if seq_ln is not None:
            a = self.seq_attention(normed, seq_ln, g)
            return self.to_out[0].tokens(a, residual=residual, **epilogue)
        if rows_ln is not None:                           # (the LayerNorm module in front, caller checked rows_ok: `normed` is the un-normalised tensor)
            qkv = ops.linear_rows(normed, self.rows_ln(rows_ln))
        elif ln is None:
            qkv = ops.conv_gemm(normed, self.fused(), ops.linear_geom(normed.shape[0]))
        else:
            qkv = ops.conv_gemm(normed, self.fused_ln(ln[0]), ops.linear_geom(normed.shape[0]), ln_stats=ln[1])
        c = self.inner
        if temporal:
            st = (g.frames * g.hw, 1, g.hw)
            a = ops.attention(qkv, 0, qkv, c, qkv, 2 * c, self.heads, g.clips, g.hw, g.frames, g.frames, st, st)
        else:
            st = (g.hw, 0, 1)
            a = ops.attention(qkv, 0, qkv, c, qkv, 2 * c, self.heads, g.images, 1, g.hw, g.hw, st, st)
        return self.to_out[0].tokens(a, residual=residual, **epilogue)

    def cross_tokens(self, normed, residual, g: Grid, kv, kv_len, ln=None, rows_ln=None, **epilogue):
        if rows_ln is not None:                           # (see self_tokens)
            q = ops.linear_rows(normed, self.rows_ln(rows_ln))
        elif ln is None:
            q = self.to_q.tokens(normed)
        else:                                             # (see self_tokens)
            q = ops.conv_gemm(normed, self.fused_ln(ln[0]), ops.linear_geom(normed.shape[0]), ln_stats=ln[1])
        a = ops.attention(q, 0, kv, 0, kv, self.inner, self.heads, g.images, 1, g.hw, kv_len,
                          (g.hw, 0, 1), (kv_len, 0, 1), kv_outer_div=g.frames)
        return self.to_out[0].tokens(a, residual=residual, **epilogue)

    def cross_tokens_temporal(self, normed, residual, g: Grid, kv, kv_len, kv_seq_mod=0):
        """Queries = the frames of one pixel, keys = a clip's context tokens (diffusers TemporalBasicTransformerBlock.attn2);
        kv_seq_mod > 0: pixel sequence number n = clip*h*w + pixel reads the context of clip n % kv_seq_mod."""
        q = self.to_q.tokens(normed)
        a = ops.attention(q, 0, kv, 0, kv, self.inner, self.heads, g.clips, g.hw, g.frames, kv_len,
                          (g.frames * g.hw, 1, g.hw), (kv_len, 0, 1), kv_seq_mod=kv_seq_mod)
        return self.to_out[0].tokens(a, residual=residual)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)
        self._pw = None
        self._pw_key = None

    def _apply(self, fn, *a, **k):
        self._pw = None
        self._pw_ln = None
        return super()._apply(fn, *a, **k)

    def packed(self):
        key = weights_key(self.proj.weight, self.proj.bias)
        if self._pw is None or self._pw_key != key:
            self._pw, self._pw_key = ops.pack_weight(self.proj.weight, self.proj.bias, geglu=True), key
        return self._pw

    def packed_ln(self, norm):
        key = weights_key(self.proj.weight, self.proj.bias, norm.weight, norm.bias)
        if self._pw_ln is None or self._pw_ln_key != key:
            self._pw_ln = ops.pack_weight(self.proj.weight, self.proj.bias, geglu=True, ln=(norm.weight, norm.bias, norm.eps))
            self._pw_ln_key = key
        return self._pw_ln

    _pw_ln = None
    _pw_ln_key = None

    def tokens(self, x, ln=None):
        if ln is None:
            return ops.conv_gemm(x, self.packed(), ops.linear_geom(x.shape[0]))
        return ops.conv_gemm(x, self.packed_ln(ln[0]), ops.linear_geom(x.shape[0]), ln_stats=ln[1])


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dim_out=None):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), Linear(dim * mult, dim_out or dim)])

    def tokens(self, x, residual, ln=None, tail=None, out=None):
        """`tail` = (packed merged weights of _MergedTail, the transformer's outer residual[, ...]): ff-out, `+ residual` and the
        transformer's proj_out as one two-source contraction (FF_PROJ_MERGE)."""
        h = self.net[0].tokens(x, ln=ln)
        if tail is None:
            return self.net[2].tokens(h, residual=residual)
        return ops.conv_gemm(h, tail[0], ops.linear_geom(h.shape[0]), x1=residual, residual=tail[1], out=out)

    def fused_tokens(self, x, norm, tail):
        """norm -> GEGLU -> ff-out -> + x -> proj_out -> + outer in ONE kernel (ops.ff_fused; tail = (merged weights, outer, ops.FFFused)); x is
        the UN-normalised residual stream.  With FF_SPLIT the rows behind the last full round of tiles take the two contractions."""
        rows = x.shape[0]
        mt, outer, pk = tail
        m = rows
        if FF_SPLIT and x.is_cuda:
            per_round = 128 * torch.cuda.get_device_properties(x.device).multi_processor_count
            full = rows // per_round * per_round
            if 0 < rows - full <= per_round // 2 and full > 0:
                m = full
        if m == rows:
            return ops.ff_fused(x, pk, outer)
        out = torch.empty_like(x)
        ops.ff_fused(x[:m], pk, outer[:m], out=out[:m])
        xr = x[m:]
        self.tokens(norm.tokens(xr), residual=xr, tail=(mt, outer[m:]), out=out[m:])
        return out


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, head_dim, cross_attention_dim=None, double_self_attention=False):
        super().__init__()
        self.norm1 = LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, head_dim)
        self.norm2 = LayerNorm(dim)
        self.attn2 = Attention(dim, None if double_self_attention else cross_attention_dim, heads, head_dim)
        self.norm3 = LayerNorm(dim)
        self.ff = FeedForward(dim)
        self.double_self_attention = double_self_attention

    def seq_pair_ok(self, x, g: Grid, temporal: bool):
        """Both attention layers on ops.seq_self_attention with the linear layers in front inside (see `tokens`)."""
        return SEQ_PRE and not self.attn2.is_cross and self.attn1.seq_ok(x, g, temporal) and self.attn2.seq_ok(x, g, temporal)

    def tokens(self, x, g: Grid, temporal: bool, text=None, text_len=0, dup: int = 1, x_stats=None, tail=None, pre_in=None):
        """`dup` > 1: x holds ONE copy of `dup` identical groups of clips (classifier-free guidance runs the same latents
        with two prompts, models/pipeline.py:165): the self-attention - which never sees the text - is computed once and
        its result replicated in front of the cross-attention; `g` describes the single copy.
        `x_stats`: ops.RowStats of x left by the contraction that produced it.  With LN_FOLD the three LayerNorms never run as
        kernels: each is folded into the projection behind it (Q|K|V, Q, GEGLU) and its row statistics come out of the epilogue
        of the projection in front of it (proj_in, to_out + residual); a producer that cannot emit them (split K, compiled
        tiles) leaves None and that LayerNorm runs as a kernel.
        `pre_in`: the transformer's proj_in (a Linear), NOT yet applied to x - it runs inside the first attention kernel (seq_pair_ok)."""
        fold = LN_FOLD
        ff_one = tail is not None and len(tail) > 2 and tail[2] is not None and ops.ff_fused_ok(x.shape[1], x.shape[0] * dup, x.dtype)
        fold_ff = fold and LN_FOLD_FF and not ff_one          # (ops.ff_fused normalises its rows itself: no statistics from the producer)
        seq1 = self.attn1.seq_ok(x, g, temporal)
        seq2 = not self.attn2.is_cross and self.attn2.seq_ok(x, g, temporal)
        if seq1 and seq2 and SEQ_PRE and dup == 1:
            # both attention layers as two kernels: [pre_in (proj_in) ->] norm1 -> attn1, then attn1.to_out + residual -> norm2 -> attn2; only
            # attn2's output projection remains a contraction (it leaves norm3's statistics)
            if pre_in is not None:
                a, x = self.attn1.seq_attention(x, self.norm1, g, pre=pre_in)
            else:
                a = self.attn1.seq_attention(x, self.norm1, g)
            a, x = self.attn2.seq_attention(a, self.norm2, g, pre=self.attn1.to_out[0], residual=x)
            r = self.attn2.to_out[0].tokens(a, residual=x, row_stats=fold_ff, coef_eps=self.norm3.eps)
            x, st = r if fold_ff else (r, None)
            if ff_one:
                return self.ff.fused_tokens(x, self.norm3, tail)
            if st is not None:
                return self.ff.tokens(x, residual=x, ln=(self.norm3, st), tail=tail)
            return self.ff.tokens(self.norm3.tokens(x), residual=x, tail=tail)
        if pre_in is not None:
            raise RuntimeError("BasicTransformerBlock: `pre_in` needs both attention layers on the fused path")
        rows1 = not seq1 and self.attn1.rows_ok(x)        # norm1 / norm2 inside the projection behind them (ops.linear_rows): no statistics from the producers
        rows2 = not seq2 and x.shape[1] == self.attn2.to_q.in_features and \
            rows_path(x.shape[1], self.attn2.inner * (1 if self.attn2.is_cross else 3), x.shape[0] * dup, x.dtype) and self.attn2.to_q.bias is None
        if seq1:                                          # norm1 + Q|K|V + attention in one kernel; norm2's statistics only if its consumer folds it
            want = fold and not seq2
            r = self.attn1.self_tokens(x, x, g, temporal, seq_ln=self.norm1, row_stats=want, coef_eps=self.norm2.eps)
            x, st = r if want else (r, None)
        elif rows1:
            want = fold and not rows2
            r = self.attn1.self_tokens(x, x, g, temporal, rows_ln=self.norm1, row_stats=want, coef_eps=self.norm2.eps)
            x, st = r if want else (r, None)
        elif fold and x_stats is not None:
            x, st = self.attn1.self_tokens(x, x, g, temporal, ln=(self.norm1, x_stats), row_stats=True, coef_eps=self.norm2.eps)
        else:
            x, st = self.attn1.self_tokens(self.norm1.tokens(x), x, g, temporal, row_stats=True, coef_eps=self.norm2.eps) if fold else \
                    (self.attn1.self_tokens(self.norm1.tokens(x), x, g, temporal), None)
        if dup > 1:
            x = torch.cat([x] * dup)
            st = None if st is None else st.repeat(dup)
            g = replace(g, clips=g.clips * dup)
        if seq2:
            r = self.attn2.self_tokens(x, x, g, temporal, seq_ln=self.norm2, row_stats=fold_ff, coef_eps=self.norm3.eps)
        else:
            rows2 = rows2 and self.attn2.rows_ok(x)
            ln2 = None if st is None or rows2 else (self.norm2, st)
            xin = x if ln2 is not None or rows2 else self.norm2.tokens(x)
            if self.attn2.is_cross:
                kv = self.attn2.text_kv(text)
                r = self.attn2.cross_tokens(xin, x, g, kv, text_len, ln=ln2, rows_ln=self.norm2 if rows2 else None, row_stats=fold_ff, coef_eps=self.norm3.eps)
            else:
                r = self.attn2.self_tokens(xin, x, g, temporal, ln=ln2, rows_ln=self.norm2 if rows2 else None, row_stats=fold_ff, coef_eps=self.norm3.eps)
        x, st = r if fold_ff else (r, None)
        if ff_one:
            return self.ff.fused_tokens(x, self.norm3, tail)
        if st is not None:
            return self.ff.tokens(x, residual=x, ln=(self.norm3, st), tail=tail)
        return self.ff.tokens(self.norm3.tokens(x), residual=x, tail=tail)


class _MergedTail:
    """Mixin of the two transformer wrappers: the packed weights of  proj_out(ff_out(h) + x)  as one contraction over [h | x]."""
    _mt = None
    _mt_key = None

    def _apply(self, fn, *a, **k):
        self._mt = None
        self._ffk = None
        return super()._apply(fn, *a, **k)

    def merged_tail(self):
        ff_out = self.transformer_blocks[-1].ff.net[2]
        w2, b2, wp, bp = ff_out.weight, ff_out.bias, self.proj_out.weight, self.proj_out.bias
        hidden, dim = w2.shape[1], w2.shape[0]
        if not FF_PROJ_MERGE or hidden % 64 or dim % 64 or wp.shape[1] != dim:       # (two-source K tiles must not straddle the sources)
            return None
        key = weights_key(w2, b2, wp, bp)
        if self._mt is None or self._mt_key != key:
            wpf = wp.detach().float()
            w = torch.cat([wpf @ w2.detach().float(), wpf], dim=1)                  # [C_out, 4C + C]
            b = (0.0 if bp is None else bp.detach().float()) + (0.0 if b2 is None else wpf @ b2.detach().float())
            b = b if torch.is_tensor(b) else None
            # The product Wp W2 is rounded ONCE to the storage type.  Its dynamic range is not that of either factor (real checkpoints,
            # LoRA-merged weights): if it leaves the type's range, or its rounding error is far above that of an ordinary weight matrix
            # of the type (half an ulp: 2^-11 fp16 / 2^-8 bf16 relative, per element; measured in the Frobenius norm), the two calls run.
            wl = w.to(wp.dtype)
            ok = bool(torch.isfinite(wl).all()) and (b is None or bool(torch.isfinite(b.to(wp.dtype)).all()))
            if ok:
                rel = ((wl.float() - w).norm() / w.norm().clamp_min(1e-30)).item()
                ok = rel <= (4.0 * 2.0 ** -11 if wp.dtype == torch.float16 else 4.0 * 2.0 ** -8)
            self._mt = ops.pack_weight(wl, None if b is None else b.to(wp.dtype)) if ok else False
            self._mt_key = key
        return self._mt or None

    _ffk = None
    _ffk_key = None

    def fused_ff(self):
        """The last block's norm3 / FeedForward and proj_out as the weight stream of ops.ff_fused (None where the library has no such kernel
        for this width or the merged weights failed merged_tail()'s rounding check)."""
        blk = self.transformer_blocks[-1]
        w1 = blk.ff.net[0].proj
        if not FF_FUSED or w1.in_features != 320 or self.merged_tail() is None:
            return None
        ff_out, n3 = blk.ff.net[2], blk.norm3
        key = weights_key(w1.weight, w1.bias, ff_out.weight, ff_out.bias, self.proj_out.weight, self.proj_out.bias, n3.weight, n3.bias)
        if self._ffk is None or self._ffk_key != key:
            self._ffk = ops.pack_ff_fused(w1.weight, w1.bias, ff_out.weight, ff_out.bias, self.proj_out.weight, self.proj_out.bias,
                                          ln=(n3.weight, n3.bias, n3.eps))
            self._ffk_key = key
        return self._ffk


class Transformer2DModel(_MergedTail, nn.Module):
    """diffusers Transformer2DModel(use_linear_projection=True)."""

    def __init__(self, heads, head_dim, in_channels, cross_attention_dim=1024, norm_num_groups=32):
        super().__init__()
        inner = heads * head_dim
        self.norm = GroupNorm(norm_num_groups, in_channels, eps=1e-6)
        self.proj_in = Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, head_dim, cross_attention_dim)])
        self.proj_out = Linear(inner, in_channels)

    def tokens(self, x, g: Grid, text, text_len, dup: int = 1):
        """`dup` > 1 (see BasicTransformerBlock.tokens): x / g are the single copy, the result covers all `dup` groups."""
        blk0 = self.transformer_blocks[0]
        want = LN_FOLD and not (blk0.attn1.to_q.bias is None and rows_path(self.proj_in.out_features, 3 * blk0.attn1.inner, x.shape[0], x.dtype))
        if not want and self.proj_in.norm_rows_ok(x, g.hw):
            h, st = self.proj_in.tokens(x, affine=(self.norm.coef_tokens(x, g.images, g.hw), g.hw)), None
        else:
            h, st = self.proj_in.tokens(self.norm.tokens(x, g.images, g.hw), row_stats=True, coef_eps=blk0.norm1.eps) if want else \
                    (self.proj_in.tokens(self.norm.tokens(x, g.images, g.hw)), None)
        outer = torch.cat([x] * dup) if dup > 1 else x
        mt, last = self.merged_tail(), len(self.transformer_blocks) - 1
        for i, blk in enumerate(self.transformer_blocks):
            h = blk.tokens(h, g, temporal=False, text=text, text_len=text_len, dup=dup if i == 0 else 1, x_stats=st if i == 0 else None,
                           tail=(mt, outer, self.fused_ff()) if mt is not None and i == last else None)
            if i == 0 and dup > 1:
                g = replace(g, clips=g.clips * dup)
        return h if mt is not None else self.proj_out.tokens(h, residual=outer)


class TransformerTemporalModel(_MergedTail, nn.Module):
    """diffusers TransformerTemporalModel(double_self_attention=True): clip-wide
    GroupNorm, then a transformer whose sequences are the T frames of one pixel (strided rows)."""

    def __init__(self, heads, head_dim, in_channels, norm_num_groups=32):
        super().__init__()
        inner = heads * head_dim
        self.norm = GroupNorm(norm_num_groups, in_channels, eps=1e-6)
        self.proj_in = Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, head_dim, None, double_self_attention=True)])
        self.proj_out = Linear(inner, in_channels)

    def tokens(self, x, g: Grid):
        blk0 = self.transformer_blocks[0]
        # (norm1's statistics are only wanted when the first attention layer folds it: ops.seq_self_attention normalises in its registers)
        want = LN_FOLD and not (SEQ_ATTN and blk0.attn1.inner in SEQ_ATTN_CHANNELS and blk0.attn1.dim_head == 64 and self.proj_in.out_features == blk0.attn1.inner
                                and ops.seq_self_attention_ok(blk0.attn1.inner, g.frames, x.shape[0], x.dtype))
        fold_norm = not want and not SEQ_PRE and self.proj_in.norm_rows_ok(x, g.frames * g.hw)
        xn = None if fold_norm else self.norm.tokens(x, g.clips, g.frames * g.hw)
        mt, last = self.merged_tail(), len(self.transformer_blocks) - 1
        if xn is not None and blk0.seq_pair_ok(xn, g, True) and self.proj_in.out_features == self.proj_in.in_features == blk0.attn1.inner:
            h = xn                                        # proj_in runs inside the first attention kernel
            for i, blk in enumerate(self.transformer_blocks):
                h = blk.tokens(h, g, temporal=True, tail=(mt, x, self.fused_ff()) if mt is not None and i == last else None, pre_in=self.proj_in if i == 0 else None)
            return h if mt is not None else self.proj_out.tokens(h, residual=x)
        if fold_norm:
            h, st = self.proj_in.tokens(x, affine=(self.norm.coef_tokens(x, g.clips, g.frames * g.hw), g.frames * g.hw)), None
        else:
            h, st = self.proj_in.tokens(xn, row_stats=True, coef_eps=blk0.norm1.eps) if want else (self.proj_in.tokens(xn), None)
        for i, blk in enumerate(self.transformer_blocks):
            h = blk.tokens(h, g, temporal=True, x_stats=st if i == 0 else None, tail=(mt, x, self.fused_ff()) if mt is not None and i == last else None)
        return h if mt is not None else self.proj_out.tokens(h, residual=x)

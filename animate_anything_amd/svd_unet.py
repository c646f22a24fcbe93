"""UNetSpatioTemporalConditionModel on MI355X: drop-in for the diffusers==0.24.0 model the reference's
Stable-Video-Diffusion path loads and calls (/root/reference/train_svd.py:85-103 - 9 input channels: mask 1 + noisy 4 +
condition 4; /root/reference/models/pipeline.py:425-431, :693-699 - the per-step call).

Same constructor arguments, `forward` signature and state-dict keys as the diffusers module; the arithmetic is the HIP token
path of `layers.py` plus the blend / frame-packing kernels of libaa_mi355.so (include/aa_mi355.h, "Stable-Video-Diffusion
path").  What the diffusers module does with permutes between [B*F,C,H,W] and [B,C,F,H,W] is addressing here: one
channels-last token matrix ordered (clip, frame, y, x) serves the per-frame and the per-clip operators alike.

  SpatioTemporalResBlock         ResnetBlock2D per frame, then GroupNorm over (C/32,F,H,W) / SiLU / Conv3d (3,1,1) twice; the
                                 learned blend alpha*x_s + (1-alpha)*(x_s + h) = x_s + (1-alpha)*h is the `acc_scale` of the
                                 second temporal conv's epilogue - no blend pass at all
  TransformerSpatioTemporalModel spatial block (self + context attention + GEGLU FF), frame-position embedding add, temporal
                                 block over the frames of one pixel (strided rows, no permute), AlphaBlender = aa_blend
  context of ONE token           (the CLIP image embedding): softmax over one key is exactly 1, so the cross-attention is a
                                 per-clip row vector added in the self-attention's output epilogue
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import AA_ACT_SILU
from .layers import (Attention, BasicTransformerBlock, Conv2d, Conv3d, Downsample2D, FeedForward, Grid, GroupNorm,
                     LayerNorm, Linear, ResnetBlock2D, TimestepEmbedding, Upsample2D, weights_key)


class UNetSpatioTemporalConditionOutput(SimpleNamespace):
    """diffusers UNetSpatioTemporalConditionOutput: `.sample` [B, F, C, H, W]."""


class AlphaBlender(nn.Module):
    """diffusers AlphaBlender.  The reference path always runs with image_only_indicator == 0 (the UNet and the VAE decoder
    build it as zeros), so `learned` and `learned_with_images` both give the scalar sigmoid(mix_factor)."""

    def __init__(self, alpha, merge_strategy="learned_with_images", switch_spatial_to_temporal_mix=False):
        super().__init__()
        if merge_strategy not in ("fixed", "learned", "learned_with_images"):
            raise ValueError(f"merge_strategy needs to be in ['learned', 'fixed', 'learned_with_images'], got {merge_strategy}")
        self.merge_strategy = merge_strategy
        self.switch_spatial_to_temporal_mix = switch_spatial_to_temporal_mix
        if merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.tensor([float(alpha)]))
        else:
            self.mix_factor = nn.Parameter(torch.tensor([float(alpha)]))
        self._alpha = None

    def alpha(self) -> float:
        """Weight of x_spatial (host float, cached per in-place version of mix_factor: reading it synchronises)."""
        key = weights_key(self.mix_factor)
        if self._alpha is None or self._alpha[0] != key:
            m = self.mix_factor.detach().float()
            a = float(m if self.merge_strategy == "fixed" else torch.sigmoid(m))
            self._alpha = (key, 1.0 - a if self.switch_spatial_to_temporal_mix else a)
        return self._alpha[1]


class TemporalResnetBlock(nn.Module):
    """diffusers TemporalResnetBlock on the (clip, frame, pixel) token grid."""

    def __init__(self, in_channels, out_channels=None, temb_channels=512, eps=1e-6):
        super().__init__()
        out_channels = out_channels or in_channels
        self.norm1 = GroupNorm(32, in_channels, eps=eps)
        self.conv1 = Conv3d(in_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = GroupNorm(32, out_channels, eps=eps)
        self.conv2 = Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))
        self.conv_shortcut = Conv3d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.tproj = None        # this block's slice of the batched time-embedding projection (set by the UNet per forward)

    def tokens(self, x, g: Grid, temb_silu=None, acc_scale=1.0):
        """x + acc_scale * branch(x): acc_scale = 1 is the diffusers block, (1 - alpha) folds the AlphaBlender behind it."""
        geom = ops.tconv_geom(g.clips, g.frames, g.hw)
        per_clip = g.frames * g.hw
        h = self.norm1.tokens(x, g.clips, per_clip, silu=True)
        if self.time_emb_proj is not None and temb_silu is not None:
            tproj = self.tproj if self.tproj is not None else self.time_emb_proj.tokens(temb_silu)
            self.tproj = None
            h = self.conv1.tokens(h, geom, rowvec=tproj, rowvec_div=per_clip)
        else:
            h = self.conv1.tokens(h, geom)
        h = self.norm2.tokens(h, g.clips, per_clip, silu=True)
        skip = x if self.conv_shortcut is None else self.conv_shortcut.tokens(x, ops.linear_geom(g.tokens))
        return self.conv2.tokens(h, geom, residual=skip, acc_scale=acc_scale)


class SpatioTemporalResBlock(nn.Module):
    """diffusers SpatioTemporalResBlock."""

    def __init__(self, in_channels, out_channels=None, temb_channels=512, eps=1e-6, temporal_eps=None, merge_factor=0.5,
                 merge_strategy="learned_with_images", switch_spatial_to_temporal_mix=False):
        super().__init__()
        out_channels = out_channels or in_channels
        self.spatial_res_block = ResnetBlock2D(in_channels, out_channels, temb_channels, eps=eps)
        self.temporal_res_block = TemporalResnetBlock(out_channels, out_channels, temb_channels,
                                                      eps=temporal_eps if temporal_eps is not None else eps)
        self.time_mixer = AlphaBlender(merge_factor, merge_strategy, switch_spatial_to_temporal_mix)

    def tokens(self, x, g: Grid, temb_silu=None, x1=None):
        xs = self.spatial_res_block.tokens(x, g, temb_silu, x1=x1)
        w_t = 1.0 - self.time_mixer.alpha()                # weight of the temporal branch: x_s + w_t * h
        if w_t == 0.0:
            self.temporal_res_block.tproj = None
            return xs
        return self.temporal_res_block.tokens(xs, g, temb_silu, acc_scale=w_t)


class SpatialTransformerBlock(BasicTransformerBlock):
    """diffusers BasicTransformerBlock as TransformerSpatioTemporalModel uses it: context attention to `text_len` tokens per clip."""

    def tokens(self, x, g: Grid, text, text_len):
        if text_len == 1:
            rv = self.attn2.cross_rowvec(text)
            x = self.attn1.self_tokens(self.norm1.tokens(x), x, g, False, rowvec=rv, rowvec_div=g.frames * g.hw)
        else:
            x = self.attn1.self_tokens(self.norm1.tokens(x), x, g, False)
            x = self.attn2.cross_tokens(self.norm2.tokens(x), x, g, self.attn2.text_kv(text), text_len)
        return self.ff.tokens(self.norm3.tokens(x), residual=x)


class TemporalBasicTransformerBlock(nn.Module):
    """diffusers TemporalBasicTransformerBlock: the sequence is the F frames of one pixel."""

    def __init__(self, dim, time_mix_inner_dim, heads, head_dim, cross_attention_dim=None):
        super().__init__()
        self.is_res = dim == time_mix_inner_dim
        self.norm_in = LayerNorm(dim)
        self.ff_in = FeedForward(dim, dim_out=time_mix_inner_dim)
        self.norm1 = LayerNorm(time_mix_inner_dim)
        self.attn1 = Attention(time_mix_inner_dim, None, heads, head_dim)
        if cross_attention_dim is not None:
            self.norm2 = LayerNorm(time_mix_inner_dim)
            self.attn2 = Attention(time_mix_inner_dim, cross_attention_dim, heads, head_dim)
        else:
            self.norm2 = self.attn2 = None
        self.norm3 = LayerNorm(time_mix_inner_dim)
        self.ff = FeedForward(time_mix_inner_dim)

    def tokens(self, x, g: Grid, text, text_len, pixel_major_context=False):
        """`pixel_major_context`: diffusers==0.24.0 hands the temporal blocks the clips' contexts broadcast as [h*w, batch]
        while their sequences are ordered [batch, h*w]: pixel sequence n = clip*h*w + pixel attends to the context of clip
        n % batch (TransformerSpatioTemporalModel.tokens)."""
        x = self.ff_in.tokens(self.norm_in.tokens(x), residual=x if self.is_res else None)
        mixed = pixel_major_context and g.clips > 1
        if self.attn2 is not None and text_len == 1 and not mixed:
            rv = self.attn2.cross_rowvec(text)
            x = self.attn1.self_tokens(self.norm1.tokens(x), x, g, True, rowvec=rv, rowvec_div=g.frames * g.hw)
        elif self.attn2 is not None and text_len == 1 and g.hw % g.clips == 0:
            x = self.attn1.self_tokens(self.norm1.tokens(x), x, g, True)
            # token row r = (image, pixel): with h*w a multiple of the batch, (clip*h*w + pixel) % batch == r % batch
            x = ops.blend(x, rowvec=self.attn2.cross_rowvec(text), rowvec_div=1, rowvec_mod=g.clips)
        else:
            x = self.attn1.self_tokens(self.norm1.tokens(x), x, g, True)
            if self.attn2 is not None:
                x = self.attn2.cross_tokens_temporal(self.norm2.tokens(x), x, g, self.attn2.text_kv(text), text_len,
                                                     kv_seq_mod=g.clips if mixed else 0)
        return self.ff.tokens(self.norm3.tokens(x), residual=x if self.is_res else None)


class TransformerSpatioTemporalModel(nn.Module):
    """diffusers TransformerSpatioTemporalModel."""

    # diffusers==0.24.0 (the version the reference pins, requirements.txt:4) builds the temporal blocks' context as
    #   time_context_first_timestep[None, :].broadcast_to(height * width, batch_size, 1, D).reshape(height * width * batch_size, 1, D)
    # i.e. ordered [pixel, batch], while TemporalBasicTransformerBlock orders its sequences [batch, pixel]: with batch > 1
    # (classifier-free guidance: [unconditional; conditional]) pixel sequence n attends to the context of clip n % batch.
    # Later diffusers releases broadcast as [batch, pixel].  True reproduces the pinned version (what the reference computes),
    # False the later, batch-aligned form.
    pixel_major_time_context = True

    def __init__(self, heads, head_dim, in_channels, num_layers=1, cross_attention_dim=None):
        super().__init__()
        if head_dim != 64:
            raise ValueError("the MI355X attention kernel implements head_dim == 64 (the SVD architecture)")
        inner = heads * head_dim
        self.in_channels = in_channels
        self.norm = GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [SpatialTransformerBlock(inner, heads, head_dim, cross_attention_dim) for _ in range(num_layers)])
        self.temporal_transformer_blocks = nn.ModuleList(
            [TemporalBasicTransformerBlock(inner, inner, heads, head_dim, cross_attention_dim) for _ in range(num_layers)])
        self.time_pos_embed = TimestepEmbedding(in_channels, in_channels * 4, out_dim=in_channels)
        self.time_mixer = AlphaBlender(0.5, "learned_with_images")
        self.proj_out = Linear(inner, in_channels)
        self._frame_idx = None

    def frame_embedding(self, frames, like):
        """time_pos_embed(Timesteps(arange(frames))) -> [frames, C]: two 14-row contractions per call, kept inside the step."""
        if self._frame_idx is None or self._frame_idx.numel() != frames or self._frame_idx.device != like.device:
            self._frame_idx = torch.arange(frames, dtype=torch.float32, device=like.device)
        return self.time_pos_embed.tokens(ops.timestep_embedding(self._frame_idx, self.in_channels, like.dtype))

    def tokens(self, x, g: Grid, text, text_len):
        h = self.proj_in.tokens(self.norm.tokens(x, g.images, g.hw))
        emb = self.frame_embedding(g.frames, x)
        a = self.time_mixer.alpha()
        for blk, tblk in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            h = blk.tokens(h, g, text, text_len)
            mix = ops.blend(h, rowvec=emb, rowvec_div=g.hw, rowvec_mod=g.frames)          # hidden_states_mix + emb
            mix = tblk.tokens(mix, g, text, text_len, self.pixel_major_time_context)
            h = ops.blend(h, mix, a, 1.0 - a)                                            # AlphaBlender
        return self.proj_out.tokens(h, residual=x)


# ----------------------------------------------------------------------------------------- UNet blocks
class _STStage(nn.Module):
    def __init__(self, res_io, temb_channels, eps, heads=None, head_dim=None, cross_attention_dim=None, layers=1,
                 down=False, up=False):
        super().__init__()
        self.has_cross_attention = heads is not None
        self.gradient_checkpointing = False
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(i, o, temb_channels, eps=eps) for i, o in res_io])
        if self.has_cross_attention:
            self.attentions = nn.ModuleList(
                [TransformerSpatioTemporalModel(heads, head_dim, o, layers, cross_attention_dim) for _, o in res_io])
        out_ch = res_io[-1][1]
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch, out_ch, padding=1)]) if down else None
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch, out_ch)]) if up else None

    def _layer(self, i, x, g, temb_silu, text, text_len, skip=None):
        x = self.resnets[i].tokens(x, g, temb_silu, x1=skip)
        if self.has_cross_attention:
            x = self.attentions[i].tokens(x, g, text, text_len)
        return x


class _DownStage(_STStage):
    def tokens(self, x, g, temb_silu, text, text_len):
        outs = []
        for i in range(len(self.resnets)):
            x = self._layer(i, x, g, temb_silu, text, text_len)
            outs.append((x, g))
        if self.downsamplers is not None:
            x, g = self.downsamplers[0].tokens(x, g)
            outs.append((x, g))
        return x, g, outs


class CrossAttnDownBlockSpatioTemporal(_DownStage):
    """diffusers CrossAttnDownBlockSpatioTemporal."""


class DownBlockSpatioTemporal(_DownStage):
    """diffusers DownBlockSpatioTemporal."""


class _UpStage(_STStage):
    def tokens(self, x, g, skips, temb_silu, text, text_len):
        for i in range(len(self.resnets)):
            skip, _ = skips.pop()
            x = self._layer(i, x, g, temb_silu, text, text_len, skip=skip)        # cat([x, skip]) is implicit
        if self.upsamplers is not None:
            x, g = self.upsamplers[0].tokens(x, g)
        return x, g


class CrossAttnUpBlockSpatioTemporal(_UpStage):
    """diffusers CrossAttnUpBlockSpatioTemporal."""


class UpBlockSpatioTemporal(_UpStage):
    """diffusers UpBlockSpatioTemporal."""


class UNetMidBlockSpatioTemporal(nn.Module):
    """diffusers UNetMidBlockSpatioTemporal."""

    def __init__(self, channels, temb_channels, heads, head_dim, cross_attention_dim, layers=1):
        super().__init__()
        self.has_cross_attention = True
        self.gradient_checkpointing = False
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(channels, channels, temb_channels, eps=1e-5) for _ in range(2)])
        self.attentions = nn.ModuleList([TransformerSpatioTemporalModel(heads, head_dim, channels, layers, cross_attention_dim)])

    def tokens(self, x, g, temb_silu, text, text_len):
        x = self.resnets[0].tokens(x, g, temb_silu)
        x = self.attentions[0].tokens(x, g, text, text_len)
        return self.resnets[1].tokens(x, g, temb_silu)


_DOWN = {"CrossAttnDownBlockSpatioTemporal": CrossAttnDownBlockSpatioTemporal, "DownBlockSpatioTemporal": DownBlockSpatioTemporal}
_UP = {"CrossAttnUpBlockSpatioTemporal": CrossAttnUpBlockSpatioTemporal, "UpBlockSpatioTemporal": UpBlockSpatioTemporal}


class UNetSpatioTemporalConditionModel(nn.Module):
    """Constructor arguments / defaults: diffusers UNetSpatioTemporalConditionModel (stable-video-diffusion-img2vid config);
    the reference instantiates it with in_channels=9 (train_svd.py:93-99).  GroupNorm epsilons per block type follow the
    diffusers blocks: 1e-6 in CrossAttnDownBlockSpatioTemporal, 1e-5 elsewhere (restated from memory)."""

    config_name = "config.json"
    _supports_gradient_checkpointing = True

    def __init__(self, sample_size=None, in_channels=8, out_channels=4,
                 down_block_types=("CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal",
                                   "CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal"),
                 up_block_types=("UpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal",
                                 "CrossAttnUpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal"),
                 block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256,
                 projection_class_embeddings_input_dim=768, layers_per_block=2, cross_attention_dim=1024,
                 transformer_layers_per_block=1, num_attention_heads=(5, 10, 20, 20), num_frames=25):
        super().__init__()
        if len(down_block_types) != len(up_block_types):
            raise ValueError(f"Must provide the same number of `down_block_types` as `up_block_types`. "
                             f"`down_block_types`: {down_block_types}. `up_block_types`: {up_block_types}.")
        if len(block_out_channels) != len(down_block_types):
            raise ValueError(f"Must provide the same number of `block_out_channels` as `down_block_types`. "
                             f"`block_out_channels`: {block_out_channels}. `down_block_types`: {down_block_types}.")
        if not isinstance(num_attention_heads, int) and len(num_attention_heads) != len(down_block_types):
            raise ValueError(f"Must provide the same number of `num_attention_heads` as `down_block_types`. "
                             f"`num_attention_heads`: {num_attention_heads}. `down_block_types`: {down_block_types}.")
        if in_channels > 16:
            raise ValueError("in_channels <= 16 is implemented (the reference uses 8 or 9)")
        n = len(block_out_channels)
        heads = (num_attention_heads,) * n if isinstance(num_attention_heads, int) else tuple(num_attention_heads)
        tl = (transformer_layers_per_block,) * n if isinstance(transformer_layers_per_block, int) \
            else tuple(transformer_layers_per_block)
        self.config = SimpleNamespace(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            block_out_channels=tuple(block_out_channels), addition_time_embed_dim=addition_time_embed_dim,
            projection_class_embeddings_input_dim=projection_class_embeddings_input_dim, layers_per_block=layers_per_block,
            cross_attention_dim=cross_attention_dim, transformer_layers_per_block=transformer_layers_per_block,
            num_attention_heads=num_attention_heads, num_frames=num_frames)
        self.sample_size = sample_size
        ch0 = block_out_channels[0]
        temb = ch0 * 4
        self.conv_in = Conv2d(in_channels, ch0, 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch0, temb)
        self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, temb)

        self.down_blocks = nn.ModuleList()
        out_c = ch0
        for i, kind in enumerate(down_block_types):
            if kind not in _DOWN:
                raise ValueError(f"{kind} does not exist.")
            in_c, out_c = out_c, block_out_channels[i]
            io = [(in_c if j == 0 else out_c, out_c) for j in range(layers_per_block)]
            cross = kind.startswith("CrossAttn")
            attn = dict(heads=heads[i], head_dim=out_c // heads[i], cross_attention_dim=cross_attention_dim, layers=tl[i]) if cross else {}
            self.down_blocks.append(_DOWN[kind](io, temb, 1e-6 if cross else 1e-5, down=i < n - 1, **attn))
        cm = block_out_channels[-1]
        self.mid_block = UNetMidBlockSpatioTemporal(cm, temb, heads[-1], cm // heads[-1], cross_attention_dim, tl[-1])
        self.up_blocks = nn.ModuleList()
        rev, rheads, rtl = list(reversed(block_out_channels)), list(reversed(heads)), list(reversed(tl))
        out_c = rev[0]
        for i, kind in enumerate(up_block_types):
            if kind not in _UP:
                raise ValueError(f"{kind} does not exist.")
            prev_c, out_c = out_c, rev[i]
            skip_c = rev[min(i + 1, n - 1)]
            L = layers_per_block + 1
            io = [((prev_c if j == 0 else out_c) + (skip_c if j == L - 1 else out_c), out_c) for j in range(L)]
            cross = kind.startswith("CrossAttn")
            attn = dict(heads=rheads[i], head_dim=out_c // rheads[i], cross_attention_dim=cross_attention_dim, layers=rtl[i]) if cross else {}
            self.up_blocks.append(_UP[kind](io, temb, 1e-5, up=i < n - 1, **attn))
        self.conv_norm_out = GroupNorm(32, ch0, eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(ch0, out_channels, 3, padding=1)
        self._graph = None
        self._packed_channels = 8 if in_channels <= 8 else 16

    # ------------------------------------------------------------------ nn.Module / diffusers protocol
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def _apply(self, fn, *a, **k):
        self.invalidate_caches()
        return super()._apply(fn, *a, **k)

    def invalidate_caches(self):
        """Drop the derived copies of the weights (batched projection packs, captured hipGraphs); see
        UNet3DConditionModel.invalidate_caches."""
        self._temb_pack = None
        self._text_pack = None
        if getattr(self, "_graph", None) is not None:
            self._graph = {}

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.invalidate_caches()
        return out

    def enable_gradient_checkpointing(self):
        return None

    def disable_gradient_checkpointing(self):
        return None

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        return None

    @classmethod
    def from_config(cls, config: dict, **overrides):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(overrides)
        import inspect
        ok = set(inspect.signature(cls.__init__).parameters) - {"self"}
        return cls(**{k: v for k, v in cfg.items() if k in ok})

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, variant=None, **overrides):
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, cls.config_name)) as f:
            model = cls.from_config(json.load(f), **overrides)
        from ._ckpt import load_state
        state = load_state(root, "diffusion_pytorch_model", variant)
        model.load_state_dict(state)
        return model.to(torch_dtype) if torch_dtype is not None else model

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(dict(vars(self.config), _class_name="UNetSpatioTemporalConditionModel"), f, indent=2)
        from safetensors.torch import save_file
        save_file({k: v.contiguous().cpu() for k, v in self.state_dict().items()},
                  os.path.join(path, "diffusion_pytorch_model.safetensors"))

    # ------------------------------------------------------------------ batched small projections
    def _project_text(self, text_tokens):
        """K | V of the context for every cross-attention layer (spatial and temporal) as ONE contraction."""
        if getattr(self, "_text_layers", None) is None:
            self._text_layers = [m for m in self.modules() if isinstance(m, Attention) and m.is_cross]
        key = weights_key(*[w_ for a in self._text_layers for w_ in (a.to_k.weight, a.to_v.weight)])
        if getattr(self, "_text_pack", None) is None or self._text_key != key:
            self._text_key = key
            layers_ = self._text_layers
            w = torch.cat([torch.cat([a.to_k.weight.detach(), a.to_v.weight.detach()], dim=0) for a in layers_], dim=0)
            offs, o = [], 0
            for a in layers_:
                offs.append(o)
                o += 2 * a.inner
            self._text_pack = (ops.pack_weight(w), layers_, offs)
        pw, layers_, offs = self._text_pack
        proj = ops.conv_gemm(text_tokens, pw, ops.linear_geom(text_tokens.shape[0]))
        for a, o in zip(layers_, offs):
            a.kv = proj[:, o:o + 2 * a.inner]

    def _project_time_embeddings(self, temb_silu):
        """time_emb_proj of every spatial and temporal resnet as ONE contraction; each block receives its column slice."""
        if getattr(self, "_temb_blocks", None) is None:
            self._temb_blocks = [m for m in self.modules()
                                 if isinstance(m, (ResnetBlock2D, TemporalResnetBlock)) and m.time_emb_proj is not None]
        key = weights_key(*[w_ for b in self._temb_blocks for w_ in (b.time_emb_proj.weight, b.time_emb_proj.bias)])
        if getattr(self, "_temb_pack", None) is None or self._temb_key != key:
            self._temb_key = key
            blocks = self._temb_blocks
            w = torch.cat([b.time_emb_proj.weight.detach() for b in blocks], dim=0)
            bias = torch.cat([b.time_emb_proj.bias.detach() for b in blocks], dim=0)
            offs, o = [], 0
            for b in blocks:
                offs.append(o)
                o += b.time_emb_proj.weight.shape[0]
            self._temb_pack = (ops.pack_weight(w, bias), blocks, offs)
        pw, blocks, offs = self._temb_pack
        proj = ops.conv_gemm(temb_silu, pw, ops.linear_geom(temb_silu.shape[0]))
        for b, o in zip(blocks, offs):
            b.tproj = proj[:, o:o + b.time_emb_proj.weight.shape[0]]

    # ------------------------------------------------------------------ hot path
    def _core(self, sources, scale, scaled_src, t, added_time_ids, text_tokens, g: Grid, text_len: int):
        """Everything between the boundary tensors: only libaa_mi355 launches (graph-capturable).
        sources: up to three [Bs, F, Cs, h, w] tensors concatenated along channels (batch element b reads b % Bs);
        scale: DEVICE fp32 scalar multiplying source `scaled_src` (scheduler.scale_model_input) or None;
        t fp32 [B]; added_time_ids fp32 [B, 3]; text_tokens [B*L, D].  Returns [B*F*h*w, out_channels] tokens."""
        dt = text_tokens.dtype
        ch0 = self.conv_in.out_channels
        cfg = self.config
        t_sin = ops.timestep_embedding(t, ch0, dt)
        emb = self.time_embedding.tokens(t_sin)
        ids_sin = ops.timestep_embedding(added_time_ids.reshape(-1), cfg.addition_time_embed_dim, dt).reshape(g.clips, -1)
        aug = self.add_embedding.tokens(ids_sin)
        temb_silu = ops.blend(emb, aug, act=AA_ACT_SILU)                              # silu(emb + aug_emb): all consumers want silu
        self._project_time_embeddings(temb_silu)
        self._project_text(text_tokens)
        x = ops.pack_frames(sources, g.clips, dt, scale, scaled_src, self._packed_channels)
        x = self.conv_in.tokens(x, ops.conv3x3_geom(g.images, g.h, g.w))
        skips = [(x, g)]
        for blk in self.down_blocks:
            x, g, outs = blk.tokens(x, g, temb_silu, text_tokens, text_len)
            skips += outs
        x = self.mid_block.tokens(x, g, temb_silu, text_tokens, text_len)
        for blk in self.up_blocks:
            x, g = blk.tokens(x, g, skips, temb_silu, text_tokens, text_len)
        x = self.conv_norm_out.tokens(x, g.images, g.hw, silu=True)
        return self.conv_out.tokens(x, ops.conv3x3_geom(g.images, g.h, g.w))

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, return_dict=True):
        """diffusers UNetSpatioTemporalConditionModel.forward: sample [B, F, C, H, W], timestep scalar or [B],
        encoder_hidden_states [B, L, D], added_time_ids [B, 3] -> .sample [B, F, out_channels, H, W]."""
        if not sample.is_cuda and not _lib.host_pointers_ok():
            raise RuntimeError("animate_anything_amd.UNetSpatioTemporalConditionModel runs on the GPU only (no CPU fallback)")
        dt, dev = self.dtype, sample.device
        b, frames, c, h, w = sample.shape
        if c != self.config.in_channels:
            raise ValueError(f"sample has {c} channels, the model expects {self.config.in_channels}")
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([float(t)], dtype=torch.float32, device=dev)
        t = t.to(device=dev, dtype=torch.float32).reshape(-1).expand(b).contiguous()
        if sample.dtype not in (torch.float32, dt):
            sample = sample.to(dt)
        sess = self.session(b, frames, h, w, tuple(encoder_hidden_states.shape[1:]), ((b, c, sample.dtype),), dev)
        sess.load(src0=sample, t=t, ids=added_time_ids.to(torch.float32), text=encoder_hidden_states)
        y = sess.run()
        y = y.reshape(b, frames, h, w, -1).permute(0, 1, 4, 2, 3)
        return UNetSpatioTemporalConditionOutput(sample=y) if return_dict else (y,)

    # ------------------------------------------------------------------ sessions: static inputs (+ hipGraph replay)
    def enable_graph(self, enabled=True):
        """Capture [embeddings + frame packing + UNet] in a hipGraph on first use per input signature and replay it afterwards."""
        self._graph = {} if enabled else None

    def session(self, batch, frames, h, w, text_shape, sources, device, scaled_src=-1):
        """Static input buffers (and, with graphs enabled, the captured hipGraph) of one input signature.
        `sources`: ((batch_i, channels_i, dtype_i), ...) of the channel-concatenated inputs."""
        key = (batch, frames, h, w, tuple(text_shape), tuple(sources), scaled_src, self.dtype, str(device))
        store = self._graph if self._graph is not None else self.__dict__.setdefault("_eager_sessions", {})
        sess = store.get(key)
        if sess is None:
            if self._graph is None:
                store.clear()
            sess = store[key] = _Session(self, key, device)
        return sess


class _Session:
    def __init__(self, net, key, device):
        b, frames, h, w, text_shape, sources, self.scaled_src, dt, _dev = key
        self.net = net
        z = lambda *s, dtype=dt: torch.zeros(*s, dtype=dtype, device=device)
        self.inputs = dict(t=z(b, dtype=torch.float32), ids=z(b, 3, dtype=torch.float32), text=z(b, *text_shape),
                           scale=torch.ones(1, dtype=torch.float32, device=device))
        for i, (sb, sc, sdt) in enumerate(sources):
            self.inputs[f"src{i}"] = z(sb, frames, sc, h, w, dtype=sdt)
        self.n_src = len(sources)
        self.text_len = text_shape[0]
        self.grid = Grid(b, frames, h, w)
        self.graph = None
        self.out = None

    def load(self, **tensors):
        for k, v in tensors.items():
            dst = self.inputs[k]
            if dst is not None and v is not None and dst.data_ptr() != v.data_ptr():
                dst.copy_(v.reshape(dst.shape) if v.numel() == dst.numel() else v.expand(dst.shape))

    def _core(self):
        i = self.inputs
        srcs = [i[f"src{k}"] for k in range(self.n_src)]
        return self.net._core(srcs, i["scale"] if self.scaled_src >= 0 else None, self.scaled_src, i["t"], i["ids"],
                              i["text"].reshape(-1, i["text"].shape[-1]), self.grid, self.text_len)

    def run(self):
        if self.net._graph is None:
            return self._core()
        if self.graph is None:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):                                   # warm-up outside capture (packs weights, autotunes)
                self._core()
            torch.cuda.current_stream().wait_stream(s)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._core()
        self.graph.replay()
        return self.out

"""Oracle for the Stable-Video-Diffusion path of the reference (BASELINE config #4): fp32 CPU restatement of

  * diffusers==0.24.0 `UNetSpatioTemporalConditionModel` and its blocks (`SpatioTemporalResBlock`,
    `TemporalResnetBlock`, `AlphaBlender`, `TransformerSpatioTemporalModel`, `TemporalBasicTransformerBlock`) - the
    model the reference loads at /root/reference/train_svd.py:85-103 (9 input channels: mask 1 + noisy 4 + condition 4),
  * diffusers `AutoencoderKLTemporalDecoder` (`TemporalDecoder`, `MidBlockTemporalDecoder`, `UpBlockTemporalDecoder`),
  * diffusers `EulerDiscreteScheduler` in the SVD configuration (v-prediction, Karras sigmas, continuous timesteps),
  * the reference's own pipelines /root/reference/models/pipeline.py:223-466 (`MaskStableVideoDiffusionPipeline`) and
    :468-731 (`TextStableVideoDiffusionPipeline`) together with the `StableVideoDiffusionPipeline` helpers they inherit.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED, like the rest of the oracle: diffusers is not
installed in the build container, the definitions below are restated from the published 0.24.0 sources; attribute names
reproduce the diffusers state-dict keys so one seeded state dict loads into this oracle and into the HIP modules.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import (Attention, BasicTransformerBlock, Downsample2D, FeedForward, ResnetBlock2D, TimestepEmbedding,
                     Upsample2D, sinusoid_embedding)
from .vae import DiagonalGaussianDistribution, Encoder


# ----------------------------------------------------------------------------------------- building blocks
class TimestepEmbeddingOut(TimestepEmbedding):
    """diffusers TimestepEmbedding(in, dim, out_dim=...): linear_2 maps to `out_dim` (TransformerSpatioTemporalModel.time_pos_embed)."""

    def __init__(self, in_channels, time_embed_dim, out_dim=None):
        super().__init__(in_channels, time_embed_dim)
        if out_dim is not None:
            self.linear_2 = nn.Linear(time_embed_dim, out_dim)


class AlphaBlender(nn.Module):
    """diffusers AlphaBlender: alpha * x_spatial + (1 - alpha) * x_temporal; `learned`: alpha = sigmoid(mix_factor);
    `learned_with_images`: alpha = 1 where image_only_indicator is set, sigmoid(mix_factor) elsewhere."""

    def __init__(self, alpha, merge_strategy="learned_with_images", switch_spatial_to_temporal_mix=False):
        super().__init__()
        self.merge_strategy = merge_strategy
        self.switch_spatial_to_temporal_mix = switch_spatial_to_temporal_mix
        if merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.tensor([float(alpha)]))
        else:
            self.mix_factor = nn.Parameter(torch.tensor([float(alpha)]))

    def get_alpha(self, image_only_indicator, ndims):
        if self.merge_strategy == "fixed":
            alpha = self.mix_factor
        elif self.merge_strategy == "learned":
            alpha = torch.sigmoid(self.mix_factor)
        else:
            alpha = torch.where(image_only_indicator.bool(), torch.ones(1, 1, device=image_only_indicator.device),
                                torch.sigmoid(self.mix_factor)[..., None])
            if ndims == 5:
                alpha = alpha[:, None, :, None, None]          # (batch, channel, frames, height, width)
            else:
                alpha = alpha.reshape(-1)[:, None, None]       # (batch*frames, height*width, channels)
        return alpha

    def forward(self, x_spatial, x_temporal, image_only_indicator=None):
        alpha = self.get_alpha(image_only_indicator, x_spatial.ndim).to(x_spatial.dtype)
        if self.switch_spatial_to_temporal_mix:
            alpha = 1.0 - alpha
        return alpha * x_spatial + (1.0 - alpha) * x_temporal


class TemporalResnetBlock(nn.Module):
    """diffusers TemporalResnetBlock: GroupNorm(32) over (C/32, T, H, W) / SiLU / Conv3d (3,1,1), twice, with the time
    embedding added per frame in between; identity (or 1x1x1 conv) shortcut."""

    def __init__(self, in_channels, out_channels=None, temb_channels=512, eps=1e-6):
        super().__init__()
        out_channels = out_channels or in_channels
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps)
        self.conv1 = nn.Conv3d(in_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps)
        self.conv2 = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))
        self.conv_shortcut = nn.Conv3d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb=None):                           # x [B, C, T, H, W], temb [B, T, Ct]
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None and temb is not None:
            t = self.time_emb_proj(F.silu(temb))[:, :, :, None, None].permute(0, 2, 1, 3, 4)
            h = h + t
        h = self.conv2(F.silu(self.norm2(h)))
        skip = x if self.conv_shortcut is None else self.conv_shortcut(x)
        return skip + h


class SpatioTemporalResBlock(nn.Module):
    """diffusers SpatioTemporalResBlock: ResnetBlock2D per frame, TemporalResnetBlock over the clip, AlphaBlender."""

    def __init__(self, in_channels, out_channels=None, temb_channels=512, eps=1e-6, temporal_eps=None, merge_factor=0.5,
                 merge_strategy="learned_with_images", switch_spatial_to_temporal_mix=False):
        super().__init__()
        out_channels = out_channels or in_channels
        self.spatial_res_block = ResnetBlock2D(in_channels, out_channels, temb_channels, eps=eps)
        self.temporal_res_block = TemporalResnetBlock(out_channels, out_channels, temb_channels,
                                                      eps=temporal_eps if temporal_eps is not None else eps)
        self.time_mixer = AlphaBlender(merge_factor, merge_strategy, switch_spatial_to_temporal_mix)

    def forward(self, x, temb=None, image_only_indicator=None):
        frames = image_only_indicator.shape[-1]
        x = self.spatial_res_block(x, temb)
        bf, c, h, w = x.shape
        b = bf // frames
        x5 = x.reshape(b, frames, c, h, w).permute(0, 2, 1, 3, 4)
        t5 = self.temporal_res_block(x5, None if temb is None else temb.reshape(b, frames, -1))
        y = self.time_mixer(x_spatial=x5, x_temporal=t5, image_only_indicator=image_only_indicator)
        return y.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class _FeedForwardOut(FeedForward):
    """diffusers FeedForward(dim, dim_out=...)."""

    def __init__(self, dim, dim_out=None, mult=4):
        super().__init__(dim, mult)
        if dim_out is not None and dim_out != dim:
            self.net[2] = nn.Linear(dim * mult, dim_out)


class TemporalBasicTransformerBlock(nn.Module):
    """diffusers TemporalBasicTransformerBlock: the frames of one pixel are the sequence."""

    def __init__(self, dim, time_mix_inner_dim, heads, head_dim, cross_attention_dim=None):
        super().__init__()
        self.is_res = dim == time_mix_inner_dim
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = _FeedForwardOut(dim, time_mix_inner_dim)
        self.norm1 = nn.LayerNorm(time_mix_inner_dim)
        self.attn1 = Attention(time_mix_inner_dim, None, heads, head_dim)
        if cross_attention_dim is not None:
            self.norm2 = nn.LayerNorm(time_mix_inner_dim)
            self.attn2 = Attention(time_mix_inner_dim, cross_attention_dim, heads, head_dim)
        else:
            self.norm2 = self.attn2 = None
        self.norm3 = nn.LayerNorm(time_mix_inner_dim)
        self.ff = FeedForward(time_mix_inner_dim)

    def forward(self, x, num_frames, encoder_hidden_states=None):
        bf, s, c = x.shape
        b = bf // num_frames
        x = x.reshape(b, num_frames, s, c).permute(0, 2, 1, 3).reshape(b * s, num_frames, c)
        res = x
        x = self.ff_in(self.norm_in(x))
        if self.is_res:
            x = x + res
        x = self.attn1(self.norm1(x)) + x
        if self.attn2 is not None:
            x = self.attn2(self.norm2(x), encoder_hidden_states) + x
        ff = self.ff(self.norm3(x))
        x = ff + x if self.is_res else ff
        return x.reshape(b, s, num_frames, c).permute(0, 2, 1, 3).reshape(bf, s, c)


class TransformerSpatioTemporalModel(nn.Module):
    """diffusers TransformerSpatioTemporalModel.  `pixel_major_time_context`: the 0.24.0 release (the reference's pin,
    requirements.txt:4) broadcasts the temporal blocks' context as [h*w, batch] although their sequences are ordered
    [batch, h*w]; later releases broadcast [batch, h*w]."""
    pixel_major_time_context = True

    def __init__(self, heads, head_dim, in_channels, num_layers=1, cross_attention_dim=None):
        super().__init__()
        inner = heads * head_dim
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, head_dim, cross_attention_dim) for _ in range(num_layers)])
        self.temporal_transformer_blocks = nn.ModuleList(
            [TemporalBasicTransformerBlock(inner, inner, heads, head_dim, cross_attention_dim) for _ in range(num_layers)])
        self.time_pos_embed = TimestepEmbeddingOut(in_channels, in_channels * 4, out_dim=in_channels)
        self.time_mixer = AlphaBlender(0.5, "learned_with_images")
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, encoder_hidden_states, image_only_indicator):
        bf, c, h, w = x.shape
        frames = image_only_indicator.shape[-1]
        b = bf // frames
        ctx = encoder_hidden_states
        first = ctx.reshape(b, frames, -1, ctx.shape[-1])[:, 0]                   # the first frame's context per clip
        if self.pixel_major_time_context:
            time_ctx = first[None].broadcast_to(h * w, b, first.shape[-2], first.shape[-1])
        else:
            time_ctx = first[:, None].broadcast_to(b, h * w, first.shape[-2], first.shape[-1])
        time_ctx = time_ctx.reshape(h * w * b, first.shape[-2], first.shape[-1])
        res = x
        t = self.norm(x).permute(0, 2, 3, 1).reshape(bf, h * w, c)
        t = self.proj_in(t)
        frame_idx = torch.arange(frames, device=x.device).repeat(b, 1).reshape(-1)
        emb = self.time_pos_embed(sinusoid_embedding(frame_idx, self.in_channels).to(t.dtype))[:, None, :]
        for blk, tblk in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            t = blk(t, ctx)
            mix = tblk(t + emb, num_frames=frames, encoder_hidden_states=time_ctx)
            t = self.time_mixer(x_spatial=t, x_temporal=mix, image_only_indicator=image_only_indicator)
        t = self.proj_out(t)
        return t.reshape(bf, h, w, c).permute(0, 3, 1, 2) + res


# ----------------------------------------------------------------------------------------- UNet blocks
class _STStage(nn.Module):
    def __init__(self, res_io, temb_channels, eps, heads=None, head_dim=None, cross_attention_dim=None, layers=1,
                 down=False, up=False):
        super().__init__()
        self.has_cross_attention = heads is not None
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(i, o, temb_channels, eps=eps) for i, o in res_io])
        if self.has_cross_attention:
            self.attentions = nn.ModuleList(
                [TransformerSpatioTemporalModel(heads, head_dim, o, layers, cross_attention_dim) for _, o in res_io])
        out_ch = res_io[-1][1]
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch, out_ch, padding=1)]) if down else None
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch, out_ch)]) if up else None

    def _layer(self, i, x, temb, ctx, ind):
        x = self.resnets[i](x, temb, ind)
        if self.has_cross_attention:
            x = self.attentions[i](x, ctx, ind)
        return x


class DownBlockST(_STStage):
    def forward(self, x, temb, ctx, ind):
        outs = []
        for i in range(len(self.resnets)):
            x = self._layer(i, x, temb, ctx, ind)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class UpBlockST(_STStage):
    def forward(self, x, skips, temb, ctx, ind):
        for i in range(len(self.resnets)):
            x = self._layer(i, torch.cat([x, skips.pop()], dim=1), temb, ctx, ind)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class MidBlockST(nn.Module):
    def __init__(self, ch, temb_channels, heads, head_dim, cross_attention_dim, layers=1):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(ch, ch, temb_channels, eps=1e-5) for _ in range(2)])
        self.attentions = nn.ModuleList([TransformerSpatioTemporalModel(heads, head_dim, ch, layers, cross_attention_dim)])

    def forward(self, x, temb, ctx, ind):
        x = self.resnets[0](x, temb, ind)
        x = self.attentions[0](x, ctx, ind)
        return self.resnets[1](x, temb, ind)


class UNetSpatioTemporalConditionModel(nn.Module):
    """diffusers UNetSpatioTemporalConditionModel (defaults = stabilityai/stable-video-diffusion-img2vid, with the
    reference's 9 input channels, train_svd.py:93-99)."""

    def __init__(self, sample_size=None, in_channels=8, out_channels=4,
                 down_block_types=("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",),
                 up_block_types=("UpBlockSpatioTemporal",) + ("CrossAttnUpBlockSpatioTemporal",) * 3,
                 block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256,
                 projection_class_embeddings_input_dim=768, layers_per_block=2, cross_attention_dim=1024,
                 transformer_layers_per_block=1, num_attention_heads=(5, 10, 20, 20), num_frames=25):
        super().__init__()
        self.config = SimpleNamespace(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                                      down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
                                      block_out_channels=tuple(block_out_channels),
                                      addition_time_embed_dim=addition_time_embed_dim,
                                      projection_class_embeddings_input_dim=projection_class_embeddings_input_dim,
                                      layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim,
                                      transformer_layers_per_block=transformer_layers_per_block,
                                      num_attention_heads=num_attention_heads, num_frames=num_frames)
        n = len(block_out_channels)
        ch0 = block_out_channels[0]
        temb = ch0 * 4
        heads = (num_attention_heads,) * n if isinstance(num_attention_heads, int) else tuple(num_attention_heads)
        tl = (transformer_layers_per_block,) * n if isinstance(transformer_layers_per_block, int) else tuple(transformer_layers_per_block)
        self.conv_in = nn.Conv2d(in_channels, ch0, 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch0, temb)
        self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, temb)
        self.down_blocks = nn.ModuleList()
        out_c = ch0
        for i, kind in enumerate(down_block_types):
            in_c, out_c = out_c, block_out_channels[i]
            io = [(in_c if j == 0 else out_c, out_c) for j in range(layers_per_block)]
            cross = kind.startswith("CrossAttn")
            attn = dict(heads=heads[i], head_dim=out_c // heads[i], cross_attention_dim=cross_attention_dim, layers=tl[i]) if cross else {}
            self.down_blocks.append(DownBlockST(io, temb, 1e-6 if cross else 1e-5, down=i < n - 1, **attn))
        cm = block_out_channels[-1]
        self.mid_block = MidBlockST(cm, temb, heads[-1], cm // heads[-1], cross_attention_dim, tl[-1])
        self.up_blocks = nn.ModuleList()
        rev, rheads, rtl = list(reversed(block_out_channels)), list(reversed(heads)), list(reversed(tl))
        out_c = rev[0]
        for i, kind in enumerate(up_block_types):
            prev_c, out_c = out_c, rev[i]
            skip_c = rev[min(i + 1, n - 1)]
            L = layers_per_block + 1
            io = [((prev_c if j == 0 else out_c) + (skip_c if j == L - 1 else out_c), out_c) for j in range(L)]
            cross = kind.startswith("CrossAttn")
            attn = dict(heads=rheads[i], head_dim=out_c // rheads[i], cross_attention_dim=cross_attention_dim, layers=rtl[i]) if cross else {}
            self.up_blocks.append(UpBlockST(io, temb, 1e-5, up=i < n - 1, **attn))
        self.conv_norm_out = nn.GroupNorm(32, ch0, eps=1e-5)
        self.conv_out = nn.Conv2d(ch0, out_channels, 3, padding=1)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, return_dict=True):
        """sample [B, F, C, H, W]; timestep scalar or [B]; encoder_hidden_states [B, L, D]; added_time_ids [B, 3]."""
        b, frames = sample.shape[:2]
        t = torch.as_tensor(timestep, dtype=torch.float32, device=sample.device).reshape(-1).expand(b)
        ch0 = self.conv_in.out_channels
        emb = self.time_embedding(sinusoid_embedding(t, ch0).to(sample.dtype))
        ids = sinusoid_embedding(added_time_ids.flatten(), self.config.addition_time_embed_dim).reshape(b, -1)
        emb = emb + self.add_embedding(ids.to(emb.dtype))
        x = sample.flatten(0, 1)
        emb = emb.repeat_interleave(frames, dim=0)
        ctx = encoder_hidden_states.repeat_interleave(frames, dim=0)
        x = self.conv_in(x)
        ind = torch.zeros(b, frames, dtype=sample.dtype, device=sample.device)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, ctx, ind)
            skips += outs
        x = self.mid_block(x, emb, ctx, ind)
        for blk in self.up_blocks:
            x = blk(x, skips, emb, ctx, ind)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        x = x.reshape(b, frames, *x.shape[1:])
        return SimpleNamespace(sample=x) if return_dict else (x,)


# ----------------------------------------------------------------------------------------- temporal-decoder VAE
class _VaeSTBlock(SpatioTemporalResBlock):
    def __init__(self, cin, cout):
        super().__init__(cin, cout, None, eps=1e-6, temporal_eps=1e-5, merge_factor=0.0, merge_strategy="learned",
                         switch_spatial_to_temporal_mix=True)


class MidBlockTemporalDecoder(nn.Module):
    def __init__(self, cin, cout, head_dim=512, num_layers=2):
        super().__init__()
        self.resnets = nn.ModuleList([_VaeSTBlock(cin if i == 0 else cout, cout) for i in range(num_layers)])
        self.attentions = nn.ModuleList([Attention(cin, None, cin // head_dim, head_dim, bias=True, norm_num_groups=32,
                                                   eps=1e-6, residual_connection=True)])

    def forward(self, x, ind):
        x = self.resnets[0](x, None, ind)
        for r, a in zip(self.resnets[1:], self.attentions):
            x = r(a(x), None, ind)
        return x


class UpBlockTemporalDecoder(nn.Module):
    def __init__(self, cin, cout, num_layers, up):
        super().__init__()
        self.resnets = nn.ModuleList([_VaeSTBlock(cin if i == 0 else cout, cout) for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout, cout)]) if up else None

    def forward(self, x, ind):
        for r in self.resnets:
            x = r(x, None, ind)
        return self.upsamplers[0](x) if self.upsamplers is not None else x


class TemporalDecoder(nn.Module):
    def __init__(self, in_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = nn.Conv2d(in_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlockTemporalDecoder(rev[0], rev[0], head_dim=rev[0], num_layers=layers_per_block)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(UpBlockTemporalDecoder(c, co, layers_per_block + 1, up=i < len(rev) - 1))
            c = co
        self.conv_norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, out_channels, 3, padding=1)
        self.time_conv_out = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, z, image_only_indicator, num_frames=1):
        x = self.conv_in(z)
        x = self.mid_block(x, image_only_indicator)
        for b in self.up_blocks:
            x = b(x, image_only_indicator)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        bf, c, h, w = x.shape
        b = bf // num_frames
        x = x.reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        x = self.time_conv_out(x)
        return x.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class AutoencoderKLTemporalDecoder(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, scaling_factor=0.18215, force_upcast=True):
        super().__init__()
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      latent_channels=latent_channels, scaling_factor=scaling_factor, force_upcast=force_upcast)
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, 32)
        self.decoder = TemporalDecoder(latent_channels, out_channels, block_out_channels, layers_per_block)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def encode(self, x):
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(self.quant_conv(self.encoder(x))))

    def decode(self, z, num_frames=1):
        b = z.shape[0] // num_frames
        ind = torch.zeros(b, num_frames, dtype=z.dtype, device=z.device)
        return SimpleNamespace(sample=self.decoder(z, ind, num_frames=num_frames))


# ----------------------------------------------------------------------------------------- scheduler
class EulerDiscreteScheduler:
    """diffusers EulerDiscreteScheduler, SVD configuration (scheduler_config.json of stable-video-diffusion-img2vid):
    scaled_linear betas 0.00085..0.012, v_prediction, leading spacing with steps_offset 1, Karras sigmas with
    sigma_min 0.002 / sigma_max 700, continuous timesteps t = 0.25 ln(sigma).  float64 numpy."""
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 prediction_type="v_prediction", timestep_spacing="leading", steps_offset=1, use_karras_sigmas=True,
                 sigma_min=0.002, sigma_max=700.0, timestep_type="continuous", rho=7.0):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type=prediction_type,
                                      timestep_spacing=timestep_spacing, steps_offset=steps_offset,
                                      use_karras_sigmas=use_karras_sigmas, sigma_min=sigma_min, sigma_max=sigma_max,
                                      timestep_type=timestep_type)
        assert beta_schedule == "scaled_linear" and use_karras_sigmas and prediction_type == "v_prediction"
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32).astype(np.float64) ** 2
        self.alphas_cumprod = np.cumprod(1.0 - betas)
        self.rho = rho
        self.sigmas = None

    def set_timesteps(self, num_inference_steps, device=None):
        c = self.config
        ramp = np.linspace(0.0, 1.0, num_inference_steps)
        lo, hi = c.sigma_min ** (1.0 / self.rho), c.sigma_max ** (1.0 / self.rho)
        sig = (hi + ramp * (lo - hi)) ** self.rho                                   # Karras et al. 2022, eq. (5)
        self.timesteps = torch.tensor(0.25 * np.log(sig), dtype=torch.float32)     # continuous v-prediction timesteps
        self.sigmas = np.concatenate([sig, [0.0]])
        self._i = 0

    @property
    def init_noise_sigma(self):
        m = float(self.sigmas.max())
        return m if self.config.timestep_spacing in ("linspace", "trailing") else (m * m + 1.0) ** 0.5

    def scale_model_input(self, sample, t=None):
        s = float(self.sigmas[self._i])
        return sample / ((s * s + 1.0) ** 0.5)

    def step(self, model_output, t, sample):
        s, sn = float(self.sigmas[self._i]), float(self.sigmas[self._i + 1])
        x0 = model_output * (-s / (s * s + 1.0) ** 0.5) + sample / (s * s + 1.0)   # v-prediction
        d = (sample - x0) / s
        self._i += 1
        return SimpleNamespace(prev_sample=sample + d * (sn - s), pred_original_sample=x0)


# ----------------------------------------------------------------------------------------- pipeline
def resize_with_antialiasing(x, size):
    """diffusers `_resize_with_antialiasing` (pipeline_stable_video_diffusion.py): Gaussian blur sized from the
    down-scaling factors, then bicubic interpolation (align_corners=True)."""
    h, w = x.shape[-2:]
    factors = (h / size[0], w / size[1])
    sigmas = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
    ks = [int(max(2.0 * 2.0 * s, 3)) for s in sigmas]
    ks = [k + 1 if k % 2 == 0 else k for k in ks]

    def kernel1d(k, s):
        xs = torch.arange(k, dtype=torch.float32) - k // 2
        if k % 2 == 0:
            xs = xs + 0.5
        g = torch.exp(-xs.pow(2.0) / (2.0 * s * s))
        return g / g.sum()

    ky, kx = kernel1d(ks[0], sigmas[0]).to(x), kernel1d(ks[1], sigmas[1]).to(x)
    c = x.shape[1]
    xp = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
    xp = F.conv2d(xp, kx.reshape(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)
    xp = F.conv2d(xp, ky.reshape(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)
    return F.interpolate(xp, size=size, mode="bicubic", align_corners=True)


def svd_pipeline(unet, vae, scheduler, image, image_embeddings, mask=None, num_frames=14, num_inference_steps=25,
                 min_guidance_scale=1.0, max_guidance_scale=3.0, fps=7, motion_bucket_id=127, noise_aug_strength=0.02,
                 decode_chunk_size=None, latents=None, aug_noise=None, condition_latent=None, output_type="latent", callback=None):
    """/root/reference/models/pipeline.py:223-466 (mask given, 9-channel UNet) and :468-731 (TextStable...: `image_embeddings`
    supplied by the caller = CLIP image embedding and/or text embedding, CFG halves already concatenated, optional
    `condition_latent`).  `image` [B,3,H,W] in [-1,1] (already pre-processed); `aug_noise` / `latents` replace the
    generator draws of the reference so the HIP pipeline can be driven with the same numbers; `callback(i, t, latents)` after
    every step (tests/test_reference_pin.py compares with the reference's own `__call__` step by step)."""
    cfg = max_guidance_scale > 1.0
    b = image.shape[0]
    decode_chunk_size = decode_chunk_size or num_frames
    fps = fps - 1
    image = image + noise_aug_strength * (torch.randn_like(image) if aug_noise is None else aug_noise)
    if condition_latent is None:
        lat = vae.encode(image).latent_dist.mode()
        if cfg:
            lat = torch.cat([torch.zeros_like(lat), lat])
        condition_latent = lat.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)
    elif cfg:
        condition_latent = torch.cat([condition_latent] * 2)
    motion_mask = unet.config.in_channels == 9
    if motion_mask:
        m = mask.reshape(1, 1, 1, *mask.shape[-2:]).to(image.dtype)
        mask5 = m.expand(2 * b if cfg else b, num_frames, 1, *mask.shape[-2:])
    ids = torch.tensor([[fps, motion_bucket_id, noise_aug_strength]], dtype=image.dtype).repeat(b, 1)
    if cfg:
        ids = torch.cat([ids, ids])
    scheduler.set_timesteps(num_inference_steps)
    h, w = image.shape[-2] // 8, image.shape[-1] // 8
    if latents is None:
        latents = torch.randn(b, num_frames, 4, h, w)
    latents = latents * scheduler.init_noise_sigma
    g = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames).reshape(1, num_frames, 1, 1, 1).to(latents.dtype)
    for i, t in enumerate(scheduler.timesteps):
        x = torch.cat([latents] * 2) if cfg else latents
        x = scheduler.scale_model_input(x, t)
        x = torch.cat([mask5, x, condition_latent], dim=2) if motion_mask else torch.cat([x, condition_latent], dim=2)
        v = unet(x, t, encoder_hidden_states=image_embeddings, added_time_ids=ids, return_dict=False)[0]
        if cfg:
            vu, vc = v.chunk(2)
            v = vu + g * (vc - vu)
        latents = scheduler.step(v, t, latents).prev_sample
        if callback is not None:
            callback(i, t, latents)
    if output_type == "latent":
        return latents
    return decode_latents(vae, latents, num_frames, decode_chunk_size)


def decode_latents(vae, latents, num_frames, decode_chunk_size=14):
    """diffusers StableVideoDiffusionPipeline.decode_latents: [B,F,C,h,w] -> [B,3,F,H,W] fp32."""
    z = latents.flatten(0, 1) / vae.config.scaling_factor
    frames = []
    for i in range(0, z.shape[0], decode_chunk_size):
        chunk = z[i:i + decode_chunk_size]
        frames.append(vae.decode(chunk, num_frames=chunk.shape[0]).sample)
    frames = torch.cat(frames)
    return frames.reshape(-1, num_frames, *frames.shape[1:]).permute(0, 2, 1, 3, 4).float()

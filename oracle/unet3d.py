"""Oracle UNet3DConditionModel: fp32 CPU restatement of
/root/reference/models/unet_3d_condition_mask.py:54-526 and the five block classes of
/root/reference/models/unet_3d_blocks.py:234-842 (inference branches only).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The module tree exposes the diffusers
state-dict keys of SURVEY.md Appendix C.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import (Downsample2D, ResnetBlock2D, TemporalConvLayer, TimestepEmbedding,
                     Transformer2DModel, TransformerTemporalModel, Upsample2D, sinusoid_embedding)


@dataclass
class UNet3DConditionOutput:
    """unet_3d_condition_mask.py:43-51."""
    sample: torch.Tensor


class _Stage(nn.Module):
    """One resolution stage.  Layer order per the reference inference branches:
    down/up with attention: resnet -> temp_conv -> spatial transformer -> temporal transformer
    (unet_3d_blocks.py:514-526, 747-759); without attention: resnet -> temp_conv (:606-609,
    :833-836).  Up stages first concatenate the popped skip tensor (:731, :828)."""

    def __init__(self, res_io, temb_channels, eps, groups, heads=None, head_dim=None,
                 cross_attention_dim=None, down=None, up=None):
        super().__init__()
        self.has_cross_attention = heads is not None
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(i, o, temb_channels, eps=eps, groups=groups) for i, o in res_io])
        self.temp_convs = nn.ModuleList([TemporalConvLayer(o, o, dropout=0.1) for _, o in res_io])
        if self.has_cross_attention:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(heads, head_dim, o, cross_attention_dim, groups) for _, o in res_io])
            self.temp_attentions = nn.ModuleList(
                [TransformerTemporalModel(heads, head_dim, o, groups) for _, o in res_io])
        out_ch = res_io[-1][1]
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch, out_ch, padding=down)]) if down is not None else None
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch, out_ch)]) if up else None

    def _layer(self, i, h, temb, text, num_frames):
        h = self.resnets[i](h, temb)
        if num_frames > 1:
            h = self.temp_convs[i](h, num_frames=num_frames)
        if self.has_cross_attention:
            h = self.attentions[i](h, encoder_hidden_states=text).sample
            if num_frames > 1:
                h = self.temp_attentions[i](h, num_frames=num_frames).sample
        return h


class _DownStage(_Stage):
    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, num_frames=1, **_):
        outs = ()
        h = hidden_states
        for i in range(len(self.resnets)):
            h = self._layer(i, h, temb, encoder_hidden_states, num_frames)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs += (h,)
        return h, outs


class CrossAttnDownBlock3D(_DownStage):
    """unet_3d_blocks.py:389-536."""


class DownBlock3D(_DownStage):
    """unet_3d_blocks.py:539-619."""


class _UpStage(_Stage):
    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None,
                upsample_size=None, num_frames=1, **_):
        h = hidden_states
        skips = list(res_hidden_states_tuple)
        for i in range(len(self.resnets)):
            h = torch.cat([h, skips.pop()], dim=1)
            h = self._layer(i, h, temb, encoder_hidden_states, num_frames)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h, upsample_size)
        return h


class CrossAttnUpBlock3D(_UpStage):
    """unet_3d_blocks.py:622-765."""


class UpBlock3D(_UpStage):
    """unet_3d_blocks.py:768-842."""


class UNetMidBlock3DCrossAttn(nn.Module):
    """unet_3d_blocks.py:234-386: resnet0 -> temp_conv0 -> [spatial attn -> temporal attn ->
    resnet -> temp_conv]."""

    def __init__(self, channels, temb_channels, eps, groups, heads, head_dim, cross_attention_dim,
                 output_scale_factor=1.0):
        super().__init__()
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([ResnetBlock2D(channels, channels, temb_channels, eps=eps, groups=groups,
                                                    output_scale_factor=output_scale_factor) for _ in range(2)])
        self.temp_convs = nn.ModuleList([TemporalConvLayer(channels, channels, dropout=0.1) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, head_dim, channels, cross_attention_dim, groups)])
        self.temp_attentions = nn.ModuleList([TransformerTemporalModel(heads, head_dim, channels, groups)])

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, num_frames=1, **_):
        # NB the reference applies temp_convs[0] unconditionally (unet_3d_blocks.py:353-354).
        h = self.resnets[0](hidden_states, temb)
        h = self.temp_convs[0](h, num_frames=num_frames)
        h = self.attentions[0](h, encoder_hidden_states=encoder_hidden_states).sample
        if num_frames > 1:
            h = self.temp_attentions[0](h, num_frames=num_frames).sample
        h = self.resnets[1](h, temb)
        if num_frames > 1:
            h = self.temp_convs[1](h, num_frames=num_frames)
        return h


class UNet3DConditionModel(nn.Module):
    """Constructor arguments and defaults: unet_3d_condition_mask.py:87-110."""

    def __init__(self, sample_size=None, in_channels=4, out_channels=4,
                 down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                 up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
                 mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
                 cross_attention_dim=1024, attention_head_dim=64, motion_mask=False, motion_strength=False):
        super().__init__()
        if len(down_block_types) != len(up_block_types):
            raise ValueError("Must provide the same number of `down_block_types` as `up_block_types`.")
        if len(block_out_channels) != len(down_block_types):
            raise ValueError("Must provide the same number of `block_out_channels` as `down_block_types`.")
        if not isinstance(attention_head_dim, int) and len(attention_head_dim) != len(down_block_types):
            raise ValueError("Must provide the same number of `attention_head_dim` as `down_block_types`.")
        self.config = SimpleNamespace(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
            downsample_padding=downsample_padding, mid_block_scale_factor=mid_block_scale_factor,
            act_fn=act_fn, norm_num_groups=norm_num_groups, norm_eps=norm_eps,
            cross_attention_dim=cross_attention_dim, attention_head_dim=attention_head_dim,
            motion_mask=motion_mask, motion_strength=motion_strength)
        self.motion_mask, self.motion_strength, self.sample_size = motion_mask, motion_strength, sample_size
        ch0 = block_out_channels[0]
        temb = ch0 * 4
        n = len(block_out_channels)
        hd = (attention_head_dim,) * n if isinstance(attention_head_dim, int) else tuple(attention_head_dim)

        self.conv_in = nn.Conv2d(in_channels, ch0, 3, padding=1)           # :137-139
        self.conv_in2 = nn.Conv2d(5, ch0, 3, padding=1)                    # :140-142
        self.time_proj_dim = ch0                                           # Timesteps(block_out_channels[0], True, 0) :146,156
        self.time_embedding = TimestepEmbedding(ch0, temb, cond_proj_dim=ch0)   # :149-154
        self.motion_embedding = nn.Sequential(nn.Linear(ch0, temb), nn.SiLU(), nn.Linear(temb, temb))  # :157-161 (dead)
        nn.init.zeros_(self.motion_embedding[-1].weight)
        nn.init.zeros_(self.motion_embedding[-1].bias)
        self.transformer_in = TransformerTemporalModel(8, hd[0], ch0, norm_num_groups)  # :163-168

        self.down_blocks = nn.ModuleList()
        out_c = ch0
        for i, kind in enumerate(down_block_types):
            in_c, out_c = out_c, block_out_channels[i]
            io = [(in_c if j == 0 else out_c, out_c) for j in range(layers_per_block)]
            down = downsample_padding if i < n - 1 else None
            if kind == "CrossAttnDownBlock3D":
                blk = CrossAttnDownBlock3D(io, temb, norm_eps, norm_num_groups, out_c // hd[i], hd[i],
                                           cross_attention_dim, down=down)
            elif kind == "DownBlock3D":
                blk = DownBlock3D(io, temb, norm_eps, norm_num_groups, down=down)
            else:
                raise ValueError(f"{kind} does not exist.")
            self.down_blocks.append(blk)

        # (registered ahead of mid_block like the reference - unet_3d_condition_mask.py:171-172,202: LoRA files address
        # layers by their position in named_modules(), so down_blocks -> up_blocks -> mid_block is part of the file format)
        self.up_blocks = nn.ModuleList()
        cm = block_out_channels[-1]
        self.mid_block = UNetMidBlock3DCrossAttn(cm, temb, norm_eps, norm_num_groups, cm // hd[-1], hd[-1],
                                                 cross_attention_dim, mid_block_scale_factor)

        rev, rhd = list(reversed(block_out_channels)), list(reversed(hd))
        out_c = rev[0]
        self.num_upsamplers = 0
        for i, kind in enumerate(up_block_types):
            prev_c, out_c = out_c, rev[i]
            skip_c = rev[min(i + 1, n - 1)]
            L = layers_per_block + 1
            io = [((prev_c if j == 0 else out_c) + (skip_c if j == L - 1 else out_c), out_c) for j in range(L)]
            up = i < n - 1
            self.num_upsamplers += int(up)
            if kind == "CrossAttnUpBlock3D":
                blk = CrossAttnUpBlock3D(io, temb, norm_eps, norm_num_groups, out_c // rhd[i], rhd[i],
                                         cross_attention_dim, up=up)
            elif kind == "UpBlock3D":
                blk = UpBlock3D(io, temb, norm_eps, norm_num_groups, up=up)
            else:
                raise ValueError(f"{kind} does not exist.")
            self.up_blocks.append(blk)

        self.conv_norm_out = nn.GroupNorm(norm_num_groups, ch0, eps=norm_eps)   # :254-258
        self.conv_out = nn.Conv2d(ch0, out_channels, 3, padding=1)              # :263-266

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, sample, timestep, encoder_hidden_states, condition_latent, mask,
                class_labels=None, timestep_cond=None, attention_mask=None, cross_attention_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                motion=None, return_dict=True, image_embeds=None):
        """unet_3d_condition_mask.py:338-526.  `attention_mask` is accepted and unused, as in the
        reference blocks (unet_3d_blocks.py:493,724)."""
        sample = torch.cat([condition_latent, sample], dim=2)                      # :376
        forward_upsample = any(s % (2 ** self.num_upsamplers) != 0 for s in sample.shape[-2:])  # :381
        b, _, num_frames = sample.shape[:3]

        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float64 if isinstance(t, float) else torch.int64)
        elif t.dim() == 0:
            t = t[None]
        t = t.to(sample.device).expand(b)
        t_emb = sinusoid_embedding(t, self.time_proj_dim).to(self.dtype)    # :408-413
        if self.motion_strength and motion is not None:
            timestep_cond = sinusoid_embedding(torch.as_tensor(motion, device=sample.device),
                                               self.time_proj_dim).to(self.dtype)   # :415
        emb = self.time_embedding(t_emb, timestep_cond)
        emb = emb.repeat_interleave(num_frames, dim=0)                              # :420
        text = encoder_hidden_states.repeat_interleave(num_frames, dim=0)           # :421

        if self.motion_mask and mask is not None:                                   # :424-428
            rep = b // mask.shape[0]
            m = mask.repeat(rep, 1, num_frames, 1, 1)      # einops 'b 1 1 h w -> (t b) 1 f h w'
            sample = torch.cat([m, sample], dim=1)
            x = sample.permute(0, 2, 1, 3, 4).reshape((b * num_frames, -1) + sample.shape[3:])
            x = self.conv_in2(x)
        else:                                                                       # :429-431
            x = sample.permute(0, 2, 1, 3, 4).reshape((b * num_frames, -1) + sample.shape[3:])
            x = self.conv_in(x)

        if num_frames > 1:
            x = self.transformer_in(x, num_frames=num_frames).sample                # :433-437

        skips = (x,)
        for blk in self.down_blocks:                                                # :440-454
            x, outs = blk(x, temb=emb, encoder_hidden_states=text, num_frames=num_frames)
            skips += outs
        if down_block_additional_residuals is not None:                             # :456-465
            skips = tuple(s + r for s, r in zip(skips, down_block_additional_residuals))

        x = self.mid_block(x, emb, encoder_hidden_states=text, num_frames=num_frames)   # :468-476
        if mid_block_additional_residual is not None:
            x = x + mid_block_additional_residual

        for i, blk in enumerate(self.up_blocks):                                    # :482-511
            k = len(blk.resnets)
            res, skips = skips[-k:], skips[:-k]
            size = skips[-1].shape[2:] if (i < len(self.up_blocks) - 1 and forward_upsample) else None
            x = blk(x, res, temb=emb, encoder_hidden_states=text, upsample_size=size, num_frames=num_frames)

        x = self.conv_out(F.silu(self.conv_norm_out(x)))                            # :514-518
        x = x.reshape((b, num_frames) + x.shape[1:]).permute(0, 2, 1, 3, 4)[:, :, 1:]   # :521-522
        return UNet3DConditionOutput(sample=x) if return_dict else (x,)

"""Layerdiffuse (RGBA video) add-ons on MI355X: drop-in for /root/reference/models/layerdiffuse_VAE.py
(`LatentTransparencyOffsetEncoder` :17-41, `UNet384` :44-177), the alpha decode of
`MaskedLatentToVideoPipeline.__call__` (/root/reference/models/pipeline_stage2.py:173-337, decode :290-318) and the
premultiplied-alpha latent preparation of /root/reference/train_transparent_i2v_stage2.py:400-426 (BASELINE.json configs[4]).

Same constructor arguments, call signatures and diffusers state-dict keys; the arithmetic is the HIP token path:
3x3 / 1x1 convolutions through aa_conv_gemm (the 64 / 128 / 256-channel levels on the LDS-DMA matrix-core kernel, the
3 / 4 / 32-channel full-resolution levels on the generic gather kernel), GroupNorm (4 groups) + SiLU through aa_groupnorm,
the 32-head x 8-channel attention of the 8x-downsampled level through aa_attention (head_dim 8).  The stage-2 denoising loop
is the stage-1 loop (same UNet3D, pipeline_stage2.py:247-288 == pipeline.py:163-198): `LatentToVideoPipeline.denoise`.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import AA_ACT_NONE, AA_ACT_SILU
from .layers import Conv2d, Downsample2D, Grid, GroupNorm, Linear, ResnetBlock2D, Upsample2D
from .pipeline import LatentToVideoPipeline, tensor2vid
from .vae import _to_tokens8


def _nchw(tokens, n, h, w):
    return tokens.reshape(n, h, w, -1).permute(0, 3, 1, 2)


def _check_device(x):
    if not x.is_cuda and not _lib.host_pointers_ok():
        raise RuntimeError("animate_anything_amd.layerdiffuse runs on the GPU only (no CPU fallback)")


class LatentTransparencyOffsetEncoder(nn.Module):
    """RGBA image [b,4,H,W] -> latent offset [b,4,H/8,W/8] (layerdiffuse_VAE.py:17-41)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        ch = [(4, 32, 1), (32, 32, 1), (32, 64, 2), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1)]
        mods = []
        for cin, cout, stride in ch:
            mods += [Conv2d(cin, cout, 3, padding=1, stride=stride), nn.SiLU()]
        last = Conv2d(256, 4, 3, padding=1)
        nn.init.zeros_(last.weight)
        nn.init.zeros_(last.bias)
        self.blocks = nn.Sequential(*mods, last)

    def forward(self, x):
        _check_device(x)
        n, _, h, w = x.shape
        dt = self.blocks[0].weight.dtype
        t = _to_tokens8(x.to(dt))
        convs = [m for m in self.blocks if isinstance(m, Conv2d)]
        for i, conv in enumerate(convs):
            geom = ops.conv3x3_geom(n, h, w, stride=conv.stride[0], pad=1)
            t = conv.tokens(t, geom, act=AA_ACT_SILU if i < len(convs) - 1 else AA_ACT_NONE)
            h, w = geom.h_out, geom.w_out
        return _nchw(t, n, h, w)

    __call__ = forward


class Attention2D(nn.Module):
    """diffusers `Attention(C, heads=C//head_dim, dim_head=head_dim, bias=True, residual_connection=True, group_norm)` as
    built by AttnDownBlock2D / UNetMidBlock2D / AttnUpBlock2D [D-0.24]: GroupNorm -> Q|K|V (one contraction) -> attention
    over the H*W tokens of an image -> to_out + residual."""

    def __init__(self, channels, head_dim, groups, eps):
        super().__init__()
        self.heads, self.head_dim = channels // head_dim, head_dim
        self.group_norm = GroupNorm(groups, channels, eps=eps)
        self.to_q = Linear(channels, channels)
        self.to_k = Linear(channels, channels)
        self.to_v = Linear(channels, channels)
        self.to_out = nn.ModuleList([Linear(channels, channels), nn.Dropout(0.0)])
        self._fused = self._fused_key = None

    def _apply(self, fn, *a, **k):
        self._fused = None
        return super()._apply(fn, *a, **k)

    def fused(self):
        from .layers import weights_key
        ws = [self.to_q.weight, self.to_k.weight, self.to_v.weight, self.to_q.bias, self.to_k.bias, self.to_v.bias]
        key = weights_key(*ws)
        if self._fused is None or self._fused_key != key:
            self._fused = ops.pack_weight(torch.cat([w.detach() for w in ws[:3]], dim=0), torch.cat([b.detach() for b in ws[3:]]))
            self._fused_key = key
        return self._fused

    def tokens(self, x, g: Grid):
        c = x.shape[1]
        qkv = ops.conv_gemm(self.group_norm.tokens(x, g.images, g.hw), self.fused(), ops.linear_geom(x.shape[0]))
        st = (g.hw, 0, 1)
        a = ops.attention(qkv, 0, qkv, c, qkv, 2 * c, self.heads, g.images, 1, g.hw, g.hw, st, st, head_dim=self.head_dim)
        return self.to_out[0].tokens(a, residual=x)


class _Block2D(nn.Module):
    """DownBlock2D / AttnDownBlock2D / UpBlock2D / AttnUpBlock2D / UNetMidBlock2D containers (temb_channels=None)."""

    def __init__(self, res_io, groups, eps, head_dim=None, n_attn=None, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(i, o, None, eps=eps, groups=groups) for i, o in res_io])
        out = res_io[-1][1]
        n_attn = len(res_io) if n_attn is None else n_attn
        self.attentions = nn.ModuleList([Attention2D(out, head_dim, groups, eps) for _ in range(n_attn)]) if head_dim else None
        self.downsamplers = nn.ModuleList([Downsample2D(out, out, padding=1)]) if down else None
        self.upsamplers = nn.ModuleList([Upsample2D(out, out)]) if up else None


class UNet384(nn.Module):
    """Alpha decoder (layerdiffuse_VAE.py:44-177): `forward(x, latent)` with x [n,3,H,W] the VAE-decoded frames and latent
    [n,4,H/8,W/8]; returns [n,4,H,W] (RGB foreground in [-1,1], alpha in [0,1])."""

    def __init__(self, in_channels=3, out_channels=4,
                 down_block_types=("DownBlock2D", "DownBlock2D", "DownBlock2D", "AttnDownBlock2D"),
                 up_block_types=("AttnUpBlock2D", "UpBlock2D", "UpBlock2D", "UpBlock2D"),
                 block_out_channels=(32, 64, 128, 256), layers_per_block=2, mid_block_scale_factor=1,
                 downsample_padding=1, downsample_type="conv", upsample_type="conv", dropout=0.0, act_fn="silu",
                 attention_head_dim=8, norm_num_groups=4, norm_eps=1e-5):
        super().__init__()
        if tuple(down_block_types) != ("DownBlock2D", "DownBlock2D", "DownBlock2D", "AttnDownBlock2D") or \
                tuple(up_block_types) != ("AttnUpBlock2D", "UpBlock2D", "UpBlock2D", "UpBlock2D") or len(block_out_channels) != 4:
            raise ValueError("UNet384: only the reference's block layout is implemented")
        if attention_head_dim != 8 or act_fn != "silu" or downsample_padding != 1 or downsample_type != "conv" or upsample_type != "conv":
            raise ValueError("UNet384: attention_head_dim=8 / silu / conv resampling (the reference's configuration) only")
        ch, g, eps, n = tuple(block_out_channels), norm_num_groups, norm_eps, 4
        self.conv_in = Conv2d(in_channels, ch[0], 3, padding=1)
        self.latent_conv_in = Conv2d(4, ch[2], 1)
        nn.init.zeros_(self.latent_conv_in.weight)
        nn.init.zeros_(self.latent_conv_in.bias)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i in range(n):
            inp, out = out, ch[i]
            io = [(inp if j == 0 else out, out) for j in range(layers_per_block)]
            self.down_blocks.append(_Block2D(io, g, eps, head_dim=attention_head_dim if i == n - 1 else None, down=i < n - 1))
        self.mid_block = _Block2D([(ch[-1], ch[-1]), (ch[-1], ch[-1])], g, eps, head_dim=attention_head_dim, n_attn=1)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(ch))
        out = rev[0]
        for i in range(n):
            prev, out = out, rev[i]
            skip_c = rev[min(i + 1, n - 1)]
            L = layers_per_block + 1
            io = [((prev if j == 0 else out) + (skip_c if j == L - 1 else out), out) for j in range(L)]
            self.up_blocks.append(_Block2D(io, g, eps, head_dim=attention_head_dim if i == 0 else None, up=i < n - 1))
        self.conv_norm_out = GroupNorm(g, ch[0], eps=eps)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(ch[0], out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def forward(self, x, latent):
        _check_device(x)
        n, _, h, w = x.shape
        dt = self.dtype
        g = Grid(1, n, h, w)
        lat8 = _to_tokens8(latent.to(dt))
        s = self.conv_in.tokens(_to_tokens8(x.to(dt)), ops.conv3x3_geom(n, h, w))
        skips = [s]
        for i, blk in enumerate(self.down_blocks):
            if i == 3:                                                          # :156-157: sample + latent_conv_in(latent)
                s = self.latent_conv_in.tokens(lat8, ops.linear_geom(lat8.shape[0]), residual=s)
            for j, r in enumerate(blk.resnets):
                s = r.tokens(s, g)
                if blk.attentions is not None:
                    s = blk.attentions[j].tokens(s, g)
                skips.append(s)
            if blk.downsamplers is not None:
                s, g = blk.downsamplers[0].tokens(s, g)
                skips.append(s)
        m = self.mid_block
        s = m.resnets[1].tokens(m.attentions[0].tokens(m.resnets[0].tokens(s, g), g), g)
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                s = r.tokens(s, g, x1=skips.pop())                              # cat([sample, skip]) is implicit
                if blk.attentions is not None:
                    s = blk.attentions[j].tokens(s, g)
            if blk.upsamplers is not None:
                s, g = blk.upsamplers[0].tokens(s, g)
        s = self.conv_norm_out.tokens(s, g.images, g.hw, silu=True)
        return _nchw(self.conv_out.tokens(s, ops.conv3x3_geom(g.images, g.h, g.w)), n, g.h, g.w)

    __call__ = forward


def encode_rgba(vae, vae_alpha_encoder, image, alpha, num_frames):
    """train_transparent_i2v_stage2.py:400-426: `image` [b,3,H,W] in [-1,1], `alpha` [b,1,H,W] in [-1,1].  The VAE encodes
    the premultiplied image, the offset encoder sees RGBA; returns the clean latents [b,4,f,h,w]."""
    a01 = (alpha + 1.0) / 2.0
    rgba = torch.cat([image, a01], dim=1)
    premul = image * a01
    lat = vae.encode(premul).latent_dist.mode() * vae.config.scaling_factor
    off = vae_alpha_encoder(rgba)
    return (lat + off.to(lat.dtype))[:, :, None].repeat(1, 1, num_frames, 1, 1)


def decode_rgba(video_tensor, latents, vae_alpha_decoder):
    """pipeline_stage2.py:290-318: returns (pngs [f,H,W,4], alpha [f,H,W], rgb [f,H,W,3]) uint8 for batch 1."""
    b, c, f, h, w = video_tensor.shape
    assert b == 1
    dt = vae_alpha_decoder.dtype
    x = video_tensor.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w).to(dt)
    lat = latents.permute(0, 2, 1, 3, 4).reshape(b * f, 4, latents.shape[-2], latents.shape[-1])
    rgba = vae_alpha_decoder(x, lat).reshape(b, f, 4, h, w).permute(0, 2, 1, 3, 4).float()
    alpha = rgba[:, 3:] * 255.0
    alpha = torch.where(alpha > 127, torch.full_like(alpha, 255.0), torch.zeros_like(alpha))
    fg = (rgba[:, :3] + 1.0) * 127.5
    pngs = torch.cat((fg, alpha), dim=1)[0].permute(1, 0, 2, 3).permute(0, 2, 3, 1)
    pngs = pngs.detach().cpu().numpy().clip(0, 255).astype(np.uint8)
    return pngs, pngs[:, :, :, 3], pngs[:, :, :, :3]


class MaskedLatentToVideoPipeline(LatentToVideoPipeline):
    """reference models/pipeline_stage2.py:173-337: the stage-1 denoising loop followed by the alpha decode."""

    @torch.no_grad()
    def __call__(self, clean_latents=None, vae_alpha_decoder=None, prompt=None, height=None, width=None, num_frames=16,
                 num_inference_steps=50, guidance_scale=9.0, negative_prompt=None, eta=0.0, generator=None, latents=None,
                 condition_latent=None, prompt_embeds=None, negative_prompt_embeds=None, output_type="np", return_dict=True,
                 callback=None, callback_steps=1, cross_attention_kwargs=None, timesteps=None, mask=None, motion=None,
                 image_embeds=None):
        if latents is None:
            raise ValueError("MaskedLatentToVideoPipeline expects caller-prepared `latents`")
        height = height or latents.shape[-2] * self.vae_scale_factor
        width = width or latents.shape[-1] * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps, negative_prompt, prompt_embeds, negative_prompt_embeds)
        do_cfg = guidance_scale > 1.0
        prompt_embeds = self._encode_prompt(prompt, latents.device, do_cfg, negative_prompt, prompt_embeds, negative_prompt_embeds)
        self.scheduler.set_timesteps(num_inference_steps, device=latents.device)
        if timesteps is None:
            timesteps = self.scheduler.timesteps
        x = self.denoise(latents, prompt_embeds, condition_latent, mask, motion, [int(t) for t in timesteps], guidance_scale,
                         callback, callback_steps)
        latents = x.to(self.unet.dtype)
        video_tensor = self.decode_latents(latents)
        pngs, alpha_jpg, pngs_rgb = decode_rgba(video_tensor, latents, vae_alpha_decoder)
        video = video_tensor if output_type == "pt" else tensor2vid(video_tensor)
        if not return_dict:
            return (video, latents, pngs, alpha_jpg, pngs_rgb)
        from .pipeline import TextToVideoSDPipelineOutput
        return TextToVideoSDPipelineOutput(frames=video)

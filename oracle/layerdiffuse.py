"""CPU oracle for the layerdiffuse (RGBA) add-ons of the reference.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates /root/reference/models/layerdiffuse_VAE.py: `LatentTransparencyOffsetEncoder` :17-41 (8 SiLU convs + a
zero-initialised conv, RGBA image -> latent offset) and `UNet384` :44-177 (diffusers 2-D UNet blocks without time embedding:
DownBlock2D x3, AttnDownBlock2D, UNetMidBlock2D, AttnUpBlock2D, UpBlock2D x3; GroupNorm groups = 4, attention_head_dim = 8;
the VAE latent enters through the zero-initialised 1x1 `latent_conv_in` in front of the fourth down block, :156-157), plus the
two call sites: the alpha decode of `MaskedLatentToVideoPipeline.__call__` (models/pipeline_stage2.py:290-318) and the
premultiplied-alpha encode of train_transparent_i2v_stage2.py:400-426.  The diffusers==0.24.0 blocks are restated from
their published definitions (un-vendored dependency): state-dict keys follow diffusers naming.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import Attention, Downsample2D, ResnetBlock2D, Upsample2D


class LatentTransparencyOffsetEncoder(nn.Module):
    """layerdiffuse_VAE.py:17-41."""

    def __init__(self):
        super().__init__()
        ch = [(4, 32, 1), (32, 32, 1), (32, 64, 2), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1)]
        mods = []
        for cin, cout, stride in ch:
            mods += [nn.Conv2d(cin, cout, 3, padding=1, stride=stride), nn.SiLU()]
        last = nn.Conv2d(256, 4, 3, padding=1)
        nn.init.zeros_(last.weight)
        nn.init.zeros_(last.bias)
        self.blocks = nn.Sequential(*mods, last)

    def forward(self, x):
        return self.blocks(x)


class _Block2D(nn.Module):
    """diffusers DownBlock2D / AttnDownBlock2D / UpBlock2D / AttnUpBlock2D with temb_channels=None [D-0.24]: resnets (each
    followed by its attention when the block has them), then the down / up sampler."""

    def __init__(self, res_io, groups, eps, head_dim=None, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(i, o, None, eps=eps, groups=groups) for i, o in res_io])
        out = res_io[-1][1]
        if head_dim:
            self.attentions = nn.ModuleList([Attention(out, None, out // head_dim, head_dim, bias=True, norm_num_groups=groups,
                                                       eps=eps, residual_connection=True) for _ in res_io])
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample2D(out, out, padding=1)]) if down else None
        self.upsamplers = nn.ModuleList([Upsample2D(out, out)]) if up else None


class UNet384(nn.Module):
    """layerdiffuse_VAE.py:44-177."""

    def __init__(self, in_channels=3, out_channels=4, block_out_channels=(32, 64, 128, 256), layers_per_block=2,
                 attention_head_dim=8, norm_num_groups=4, norm_eps=1e-5):
        super().__init__()
        ch, g, eps, n = block_out_channels, norm_num_groups, norm_eps, len(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.latent_conv_in = nn.Conv2d(4, ch[2], 1)
        nn.init.zeros_(self.latent_conv_in.weight)
        nn.init.zeros_(self.latent_conv_in.bias)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i in range(n):
            inp, out = out, ch[i]
            io = [(inp if j == 0 else out, out) for j in range(layers_per_block)]
            self.down_blocks.append(_Block2D(io, g, eps, head_dim=attention_head_dim if i == n - 1 else None, down=i < n - 1))
        self.mid_block = _Block2D([(ch[-1], ch[-1]), (ch[-1], ch[-1])], g, eps)
        self.mid_block.attentions = nn.ModuleList([Attention(ch[-1], None, ch[-1] // attention_head_dim, attention_head_dim, bias=True,
                                                             norm_num_groups=g, eps=eps, residual_connection=True)])
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(ch))
        out = rev[0]
        for i in range(n):
            prev, out = out, rev[i]
            skip_c = rev[min(i + 1, n - 1)]
            L = layers_per_block + 1
            io = [((prev if j == 0 else out) + (skip_c if j == L - 1 else out), out) for j in range(L)]
            self.up_blocks.append(_Block2D(io, g, eps, head_dim=attention_head_dim if i == 0 else None, up=i < n - 1))
        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(ch[0], out_channels, 3, padding=1)

    def forward(self, x, latent):
        sample_latent = self.latent_conv_in(latent)
        sample = self.conv_in(x)
        skips = [sample]
        for i, blk in enumerate(self.down_blocks):
            if i == 3:
                sample = sample + sample_latent                              # :156-157
            for j, r in enumerate(blk.resnets):
                sample = r(sample)
                if blk.attentions is not None:
                    sample = blk.attentions[j](sample)
                skips.append(sample)
            if blk.downsamplers is not None:
                sample = blk.downsamplers[0](sample)
                skips.append(sample)
        m = self.mid_block
        sample = m.resnets[1](m.attentions[0](m.resnets[0](sample)))
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                sample = r(torch.cat([sample, skips.pop()], dim=1))
                if blk.attentions is not None:
                    sample = blk.attentions[j](sample)
            if blk.upsamplers is not None:
                sample = blk.upsamplers[0](sample)
        return self.conv_out(F.silu(self.conv_norm_out(sample)))


def encode_rgba(vae, alpha_encoder, image, alpha, num_frames):
    """train_transparent_i2v_stage2.py:400-426: `image` [b,3,H,W] in [-1,1], `alpha` [b,1,H,W] in [-1,1]; the VAE encodes
    the PREMULTIPLIED image, the offset encoder the RGBA image; returns latents [b,4,f,h,w] = repeat(vae latent + offset)."""
    a01 = (alpha + 1.0) / 2.0
    rgba = torch.cat([image, a01], dim=1)
    premul = image * a01
    lat = vae.encode(premul).latent_dist.mode() * vae.config.scaling_factor
    off = alpha_encoder(rgba)
    return (lat + off)[:, :, None].repeat(1, 1, num_frames, 1, 1)


def decode_rgba(video_tensor, latents, alpha_decoder):
    """models/pipeline_stage2.py:290-318: `video_tensor` [b,3,f,H,W] (VAE decode of the latents), `latents` [b,4,f,h,w];
    returns (pngs [f,H,W,4] uint8, alpha [f,H,W] uint8, rgb [f,H,W,3] uint8)."""
    b, c, f, h, w = video_tensor.shape
    x = video_tensor.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    lat = latents.permute(0, 2, 1, 3, 4).reshape(b * f, 4, latents.shape[-2], latents.shape[-1])
    rgba = alpha_decoder(x, lat).reshape(b, f, 4, h, w).permute(0, 2, 1, 3, 4)
    alpha = rgba[:, 3:] * 255.0
    alpha = torch.where(alpha > 127, torch.full_like(alpha, 255.0), torch.zeros_like(alpha))
    fg = (rgba[:, :3] + 1.0) * 127.5
    pngs = torch.cat((fg, alpha), dim=1)[0].permute(1, 0, 2, 3).permute(0, 2, 3, 1)
    pngs = pngs.detach().cpu().float().numpy().clip(0, 255).astype(np.uint8)
    return pngs, pngs[:, :, :, 3], pngs[:, :, :, :3]

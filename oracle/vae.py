"""Oracle AutoencoderKL: fp32 CPU restatement of diffusers==0.24.0 `AutoencoderKL` in the SD /
ModelScope configuration.  Reference call sites:
/root/reference/utils/common.py:12-20 (encode -> latent_dist.mode()), /root/reference/train.py:89,847
(load, enable_slicing) and /root/reference/models/pipeline.py:200 (decode_latents).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import Attention, Downsample2D, ResnetBlock2D, Upsample2D


class _MidBlock(nn.Module):
    def __init__(self, ch, groups, eps=1e-6, head_dim=None):
        super().__init__()
        head_dim = head_dim or ch
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, None, eps=eps, groups=groups) for _ in range(2)])
        self.attentions = nn.ModuleList([Attention(ch, None, ch // head_dim, head_dim, bias=True,
                                                   norm_num_groups=groups, eps=eps, residual_connection=True)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, eps=1e-6, groups=groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, cout, padding=0)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.downsamplers[0](x) if self.downsamplers is not None else x


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, eps=1e-6, groups=groups) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout, cout)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.upsamplers[0](x) if self.upsamplers is not None else x


class Encoder(nn.Module):
    def __init__(self, in_channels, latent_channels, chans, layers, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, chans[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = chans[0]
        for i, co in enumerate(chans):
            self.down_blocks.append(_EncBlock(c, co, layers, groups, down=i < len(chans) - 1))
            c = co
        self.mid_block = _MidBlock(c, groups)
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, latent_channels, out_channels, chans, layers, groups):
        super().__init__()
        rev = list(reversed(chans))
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = _MidBlock(rev[0], groups)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(_DecBlock(c, co, layers + 1, groups, up=i < len(rev) - 1))
            c = co
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DiagonalGaussianDistribution:
    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype, device=self.mean.device)
        return self.mean + self.std * noise


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, scaling_factor=0.18215, force_upcast=True):
        super().__init__()
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      latent_channels=latent_channels, norm_num_groups=norm_num_groups,
                                      scaling_factor=scaling_factor, force_upcast=force_upcast)
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.use_slicing = False

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def _sliced(self, fn, x):
        if self.use_slicing and x.shape[0] > 1:
            return torch.cat([fn(s) for s in x.split(1)])
        return fn(x)

    def encode(self, x):
        moments = self._sliced(lambda s: self.quant_conv(self.encoder(s)), x)
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(moments))

    def decode(self, z):
        return SimpleNamespace(sample=self._sliced(lambda s: self.decoder(self.post_quant_conv(s)), z))

"""AutoencoderKLTemporalDecoder on MI355X: drop-in for the diffusers==0.24.0 VAE of the Stable-Video-Diffusion path
(loaded with the pipeline at /root/reference/train_svd.py:85-91; `_encode_vae_image` / `decode_latents` of diffusers'
StableVideoDiffusionPipeline, called from /root/reference/models/pipeline.py:377,453 and :649,718).

The encoder is the AutoencoderKL encoder (vae.py).  The decoder's SpatioTemporalResBlocks (no time embedding, `learned`
blend with switch_spatial_to_temporal_mix) reuse the UNet's token-path blocks: per-frame ResnetBlock2D, then the temporal
GroupNorm / Conv3d (3,1,1) pair whose second conv applies the blend weight in its epilogue.  `time_conv_out` (Conv3d 3->3
over the frames) reads the 3-channel image padded to 8 channels.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib, ops
from .layers import Conv2d, Conv3d, Grid, GroupNorm, Upsample2D
from .svd_unet import SpatioTemporalResBlock
from .vae import DiagonalGaussianDistribution, Encoder, VaeAttention, _to_tokens8


def _st_block(cin, cout):
    return SpatioTemporalResBlock(cin, cout, None, eps=1e-6, temporal_eps=1e-5, merge_factor=0.0, merge_strategy="learned",
                                  switch_spatial_to_temporal_mix=True)


class MidBlockTemporalDecoder(nn.Module):
    """diffusers MidBlockTemporalDecoder."""

    def __init__(self, cin, cout, num_layers=2):
        super().__init__()
        self.resnets = nn.ModuleList([_st_block(cin if i == 0 else cout, cout) for i in range(num_layers)])
        self.attentions = nn.ModuleList([VaeAttention(cin, 32, eps=1e-6)])

    def tokens(self, x, g):
        x = self.resnets[0].tokens(x, g)
        for r, a in zip(self.resnets[1:], self.attentions):
            x = r.tokens(a.tokens(x, g), g)
        return x


class UpBlockTemporalDecoder(nn.Module):
    """diffusers UpBlockTemporalDecoder."""

    def __init__(self, cin, cout, num_layers, up):
        super().__init__()
        self.resnets = nn.ModuleList([_st_block(cin if i == 0 else cout, cout) for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout, cout)]) if up else None

    def tokens(self, x, g):
        for r in self.resnets:
            x = r.tokens(x, g)
        if self.upsamplers is not None:
            x, g = self.upsamplers[0].tokens(x, g)
        return x, g


class TemporalDecoder(nn.Module):
    """diffusers TemporalDecoder."""

    def __init__(self, in_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = Conv2d(in_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlockTemporalDecoder(rev[0], rev[0], num_layers=layers_per_block)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(UpBlockTemporalDecoder(c, co, layers_per_block + 1, up=i < len(rev) - 1))
            c = co
        self.conv_norm_out = GroupNorm(32, c, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(c, out_channels, 3, padding=1)
        self.time_conv_out = Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))

    def tokens(self, z8, g: Grid):
        """z8 [clips*frames*h*w, 8] -> ([clips*frames*8h*8w, out_channels], grid)."""
        x = self.conv_in.tokens(z8, ops.conv3x3_geom(g.images, g.h, g.w))
        x = self.mid_block.tokens(x, g)
        for b in self.up_blocks:
            x, g = b.tokens(x, g)
        x = self.conv_norm_out.tokens(x, g.images, g.hw, silu=True)
        img8 = torch.zeros(g.tokens, 8, dtype=x.dtype, device=x.device)
        self.conv_out.tokens(x, ops.conv3x3_geom(g.images, g.h, g.w), out=img8)      # out_channels of 8 columns written
        return self.time_conv_out.tokens(img8, ops.tconv_geom(g.clips, g.frames, g.hw)), g


class AutoencoderKLTemporalDecoder(nn.Module):
    config_name = "config.json"

    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, scaling_factor=0.18215, force_upcast=True, **_):
        super().__init__()
        if in_channels > 8 or out_channels > 8 or latent_channels > 4:
            raise ValueError("AutoencoderKLTemporalDecoder: in/out_channels <= 8 and latent_channels <= 4 are implemented")
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      latent_channels=latent_channels, scaling_factor=scaling_factor, force_upcast=force_upcast)
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, 32)
        self.decoder = TemporalDecoder(latent_channels, out_channels, block_out_channels, layers_per_block)
        self.quant_conv = Conv2d(2 * latent_channels, 2 * latent_channels, 1)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, variant=None, **overrides):
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, cls.config_name)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        cfg.update(overrides)
        import inspect
        ok = set(inspect.signature(cls.__init__).parameters) - {"self", "_"}
        model = cls(**{k: v for k, v in cfg.items() if k in ok})
        from ._ckpt import load_state
        state = load_state(root, "diffusion_pytorch_model", variant)
        model.load_state_dict(state)
        return model.to(torch_dtype) if torch_dtype is not None else model

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(dict(vars(self.config), _class_name="AutoencoderKLTemporalDecoder"), f, indent=2)
        from safetensors.torch import save_file
        save_file({k: v.contiguous().cpu() for k, v in self.state_dict().items()},
                  os.path.join(path, "diffusion_pytorch_model.safetensors"))

    def _guard(self, x):
        if not x.is_cuda and not _lib.host_pointers_ok():
            raise RuntimeError("animate_anything_amd.AutoencoderKLTemporalDecoder runs on the GPU only (no CPU fallback)")

    def encode(self, x):
        """x [N,3,H,W] in [-1,1] -> .latent_dist (DiagonalGaussianDistribution over [N,4,H/8,W/8])."""
        self._guard(x)
        x = x.to(self.dtype)
        n, _, h, w = x.shape
        m, g = self.encoder.tokens(_to_tokens8(x), Grid(1, n, h, w))
        m = self.quant_conv.tokens(m, ops.linear_geom(m.shape[0]))
        moments = m.reshape(n, g.h, g.w, -1).permute(0, 3, 1, 2).contiguous()
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(moments))

    def decode(self, z, num_frames=1):
        """z [B*F,4,h,w], F = num_frames consecutive frames per clip -> .sample [B*F,3,8h,8w]."""
        self._guard(z)
        z = z.to(self.dtype)
        n, _, h, w = z.shape
        if n % num_frames:
            raise ValueError(f"decode: {n} latents are not a whole number of {num_frames}-frame clips")
        y, g = self.decoder.tokens(_to_tokens8(z), Grid(n // num_frames, num_frames, h, w))
        img = y.reshape(n, g.h, g.w, -1).permute(0, 3, 1, 2).contiguous()
        return SimpleNamespace(sample=img)

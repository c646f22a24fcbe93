"""AutoencoderKL on MI355X: drop-in for the diffusers==0.24.0 `AutoencoderKL` the reference wraps
around the denoiser (load: /root/reference/train.py:89; encode: /root/reference/utils/common.py:12-20;
decode: `decode_latents` at /root/reference/models/pipeline.py:200; `enable_slicing`: train.py:847).
SD / ModelScope configuration; same state-dict keys.

All convolutions / linears are the implicit-GEMM kernel, GroupNorm(+SiLU) the fused norm kernel.
The single-head, head_dim-512 mid-block attention materialises its score matrix per image
(fp32 scores -> row softmax -> P V), three launches of the same contraction kernel plus a softmax.
Frames are processed as one batch: `enable_slicing()` is accepted (results are identical, there is
no memory reason to slice on a 288 GB part).
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib, ops
from .layers import Conv2d, Downsample2D, Grid, GroupNorm, Linear, ResnetBlock2D, Upsample2D


class VaeAttention(nn.Module):
    """diffusers `Attention(heads=1, dim_head=C, bias=True, residual_connection=True, group_norm)`
    as used by the VAE `UNetMidBlock2D`."""

    def __init__(self, channels, groups, eps=1e-6):
        super().__init__()
        self.group_norm = GroupNorm(groups, channels, eps=eps)
        self.to_q = Linear(channels, channels)
        self.to_k = Linear(channels, channels)
        self.to_v = Linear(channels, channels)
        self.to_out = nn.ModuleList([Linear(channels, channels), nn.Dropout(0.0)])

    def tokens(self, x, g: Grid):
        c, hw, n = x.shape[1], g.hw, g.images
        dt, dev = x.dtype, x.device
        kp = (hw + 63) // 64 * 64
        slack = 128
        gn = torch.zeros(n * hw + slack, c, dtype=dt, device=dev)
        self.group_norm.tokens(x, n, hw, out=gn)
        gn_rows = gn[: n * hw]
        q = self.to_q.tokens(gn_rows)
        k = torch.zeros(n * hw + slack, c, dtype=dt, device=dev)
        self.to_k.tokens(gn_rows, out=k)
        wv = self.to_v.weight.detach().contiguous()                       # [C, C]: rows act as "tokens"
        attn = torch.empty(n * hw, c, dtype=dt, device=dev)
        scores = torch.empty(hw, kp, dtype=torch.float32, device=dev)
        probs = torch.zeros(hw, kp, dtype=dt, device=dev)
        v_t = torch.zeros(c, kp, dtype=dt, device=dev)
        for i in range(n):
            rows = slice(i * hw, (i + 1) * hw)
            k_i = ops.PackedWeight(k[i * hw: i * hw + kp], None, hw, 1, 1, c)
            ops.conv_gemm(q[rows], k_i, ops.linear_geom(hw), out=scores, out_dtype=torch.float32,
                          out_scale=float(c) ** -0.5, bias=None)
            ops.softmax_rows(scores, dt, cols=hw, out=probs)
            # V^T[c, pos] = Wv[c, :] . gn[pos, :] + bv[c]  (bias per output row)
            gn_i = ops.PackedWeight(gn[i * hw: i * hw + kp], None, hw, 1, 1, c)
            ops.conv_gemm(wv, gn_i, ops.linear_geom(c), out=v_t, bias=self.to_v.bias, bias_per_row=True)
            ops.conv_gemm(probs, ops.PackedWeight(v_t, None, c, 1, 1, kp), ops.linear_geom(hw), out=attn[rows], bias=None)
        return self.to_out[0].tokens(attn, residual=x)


class _MidBlock(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, None, eps=1e-6, groups=groups) for _ in range(2)])
        self.attentions = nn.ModuleList([VaeAttention(ch, groups)])

    def tokens(self, x, g):
        x = self.resnets[0].tokens(x, g)
        x = self.attentions[0].tokens(x, g)
        return self.resnets[1].tokens(x, g)


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, eps=1e-6, groups=groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, cout, padding=0)]) if down else None

    def tokens(self, x, g):
        for r in self.resnets:
            x = r.tokens(x, g)
        if self.downsamplers is not None:
            x, g = self.downsamplers[0].tokens(x, g)
        return x, g


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, eps=1e-6, groups=groups) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout, cout)]) if up else None

    def tokens(self, x, g):
        for r in self.resnets:
            x = r.tokens(x, g)
        if self.upsamplers is not None:
            x, g = self.upsamplers[0].tokens(x, g)
        return x, g


class Encoder(nn.Module):
    def __init__(self, in_channels, latent_channels, chans, layers, groups):
        super().__init__()
        self.conv_in = Conv2d(in_channels, chans[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = chans[0]
        for i, co in enumerate(chans):
            self.down_blocks.append(_EncBlock(c, co, layers, groups, down=i < len(chans) - 1))
            c = co
        self.mid_block = _MidBlock(c, groups)
        self.conv_norm_out = GroupNorm(groups, c, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(c, 2 * latent_channels, 3, padding=1)

    def tokens(self, x, g):
        x = self.conv_in.tokens(x, ops.conv3x3_geom(g.images, g.h, g.w))
        for b in self.down_blocks:
            x, g = b.tokens(x, g)
        x = self.mid_block.tokens(x, g)
        x = self.conv_norm_out.tokens(x, g.images, g.hw, silu=True)
        return self.conv_out.tokens(x, ops.conv3x3_geom(g.images, g.h, g.w)), g


class Decoder(nn.Module):
    def __init__(self, latent_channels, out_channels, chans, layers, groups):
        super().__init__()
        rev = list(reversed(chans))
        self.conv_in = Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = _MidBlock(rev[0], groups)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(_DecBlock(c, co, layers + 1, groups, up=i < len(rev) - 1))
            c = co
        self.conv_norm_out = GroupNorm(groups, c, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(c, out_channels, 3, padding=1)

    def tokens(self, z8, g):
        x = self.conv_in.tokens(z8, ops.conv3x3_geom(g.images, g.h, g.w))
        x = self.mid_block.tokens(x, g)
        for b in self.up_blocks:
            x, g = b.tokens(x, g)
        x = self.conv_norm_out.tokens(x, g.images, g.hw, silu=True)
        return self.conv_out.tokens(x, ops.conv3x3_geom(g.images, g.h, g.w)), g


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


def _to_tokens8(x):
    """[N,C,H,W] (C <= 8) -> channels-last tokens zero-padded to 8 channels."""
    n, c, h, w = x.shape
    t = torch.zeros(n, h, w, 8, dtype=x.dtype, device=x.device)
    t[..., :c] = x.permute(0, 2, 3, 1)
    return t.reshape(-1, 8)


class AutoencoderKL(nn.Module):
    config_name = "config.json"

    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, scaling_factor=0.18215, force_upcast=True, **_):
        super().__init__()
        if in_channels > 8 or latent_channels > 4:
            raise ValueError("AutoencoderKL: in_channels <= 8 and latent_channels <= 4 are implemented")
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      latent_channels=latent_channels, norm_num_groups=norm_num_groups,
                                      scaling_factor=scaling_factor, force_upcast=force_upcast)
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = Conv2d(latent_channels, latent_channels, 1)
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
            json.dump(dict(vars(self.config), _class_name="AutoencoderKL"), f, indent=2)
        from safetensors.torch import save_file
        save_file({k: v.contiguous().cpu() for k, v in self.state_dict().items()},
                  os.path.join(path, "diffusion_pytorch_model.safetensors"))

    def _guard(self, x):
        if not x.is_cuda and not _lib.host_pointers_ok():
            raise RuntimeError("animate_anything_amd.AutoencoderKL runs on the GPU only (no CPU fallback)")

    def encode(self, x):
        """x [N,3,H,W] in [-1,1] -> .latent_dist (DiagonalGaussianDistribution over [N,4,H/8,W/8])."""
        self._guard(x)
        x = x.to(self.dtype)
        n, _, h, w = x.shape
        m, g = self.encoder.tokens(_to_tokens8(x), Grid(1, n, h, w))
        m = self.quant_conv.tokens(m, ops.linear_geom(m.shape[0]))
        moments = m.reshape(n, g.h, g.w, -1).permute(0, 3, 1, 2).contiguous()
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(moments))

    def decode(self, z):
        """z [N,4,h,w] -> .sample [N,3,8h,8w]."""
        self._guard(z)
        z = z.to(self.dtype)
        n, _, h, w = z.shape
        z8 = _to_tokens8(z)
        pq = torch.zeros_like(z8)
        self.post_quant_conv.tokens(z8, ops.linear_geom(z8.shape[0]), out=pq)      # 4 of 8 columns written
        y, g = self.decoder.tokens(pq, Grid(1, n, h, w))
        img = y.reshape(n, g.h, g.w, -1).permute(0, 3, 1, 2).contiguous()
        return SimpleNamespace(sample=img)

"""Timesteps / TimestepEmbedding -> oracle.layers (reference call sites: unet_3d_condition_mask.py:146-161,408-421)."""
import torch

from oracle import layers as L


class Timesteps(torch.nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        return L.sinusoid_embedding(timesteps, self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift)


class TimestepEmbedding(L.TimestepEmbedding):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None):
        assert act_fn == "silu" and out_dim is None and post_act_fn is None
        super().__init__(in_channels, time_embed_dim, cond_proj_dim)

"""Stand-ins for the diffusers-0.24 2-D UNet blocks that the reference's `UNet384` builds (models/layerdiffuse_VAE.py:7,
44-177: `get_down_block` / `UNetMidBlock2D` / `get_up_block` with temb_channels=None), on top of oracle leaves and with the
diffusers call signatures, so that the reference's OWN constructor arithmetic and `forward` (latent injection in front of the
fourth down block :156-157, the residual-tuple bookkeeping :166-169) run here.  The block bodies below restate the published
0.24.0 source (DownBlock2D, AttnDownBlock2D, UNetMidBlock2D, UpBlock2D, AttnUpBlock2D) and are therefore not pinned by running
them.  TEST INFRASTRUCTURE ONLY."""
import torch
import torch.nn as nn

from oracle.layers import Attention, Downsample2D, ResnetBlock2D, Upsample2D


def _attn(ch, head_dim, groups, eps):
    return Attention(ch, None, ch // head_dim, head_dim, bias=True, norm_num_groups=groups, eps=eps, residual_connection=True)


class _Down(nn.Module):
    def __init__(self, num_layers, in_channels, out_channels, add_downsample, resnet_eps, resnet_groups, head_dim=None, downsample_padding=1):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, None, eps=resnet_eps, groups=resnet_groups)
                                      for i in range(num_layers)])
        self.attentions = nn.ModuleList([_attn(out_channels, head_dim, resnet_groups, resnet_eps) for _ in range(num_layers)]) if head_dim else None
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, out_channels, padding=downsample_padding)]) if add_downsample else None

    def forward(self, hidden_states, temb=None):
        outs = ()
        for i, r in enumerate(self.resnets):
            hidden_states = r(hidden_states, temb)
            if self.attentions is not None:
                hidden_states = self.attentions[i](hidden_states)
            outs += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            outs += (hidden_states,)
        return hidden_states, outs


class _Up(nn.Module):
    def __init__(self, num_layers, in_channels, out_channels, prev_output_channel, add_upsample, resnet_eps, resnet_groups, head_dim=None):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            cin = prev_output_channel if i == 0 else out_channels
            res.append(ResnetBlock2D(cin + skip, out_channels, None, eps=resnet_eps, groups=resnet_groups))
        self.resnets = nn.ModuleList(res)
        self.attentions = nn.ModuleList([_attn(out_channels, head_dim, resnet_groups, resnet_eps) for _ in range(num_layers)]) if head_dim else None
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None):
        for i, r in enumerate(self.resnets):
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = r(torch.cat([hidden_states, res], dim=1), temb)
            if self.attentions is not None:
                hidden_states = self.attentions[i](hidden_states)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states)
        return hidden_states


class UNetMidBlock2D(nn.Module):
    def __init__(self, in_channels, temb_channels=None, dropout=0.0, num_layers=1, resnet_eps=1e-6, resnet_time_scale_shift="default",
                 resnet_act_fn="swish", resnet_groups=32, attn_groups=None, resnet_pre_norm=True, add_attention=True,
                 attention_head_dim=1, output_scale_factor=1.0):
        super().__init__()
        assert temb_channels is None and resnet_time_scale_shift == "default" and add_attention and output_scale_factor == 1
        groups = attn_groups if attn_groups is not None else resnet_groups            # (0.24: attn_groups=None -> resnet_groups)
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels, in_channels, None, eps=resnet_eps, groups=resnet_groups) for _ in range(num_layers + 1)])
        self.attentions = nn.ModuleList([_attn(in_channels, attention_head_dim, groups, resnet_eps) for _ in range(num_layers)])

    def forward(self, hidden_states, temb=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        for attn, r in zip(self.attentions, self.resnets[1:]):
            hidden_states = r(attn(hidden_states), temb)
        return hidden_states


def get_down_block(down_block_type, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps, resnet_act_fn,
                   resnet_groups=None, attention_head_dim=None, downsample_padding=None, resnet_time_scale_shift="default",
                   downsample_type=None, dropout=0.0, **unused):
    assert temb_channels is None and resnet_act_fn == "silu" and resnet_time_scale_shift == "default" and dropout == 0.0
    if down_block_type == "DownBlock2D":
        return _Down(num_layers, in_channels, out_channels, add_downsample, resnet_eps, resnet_groups, None, downsample_padding)
    if down_block_type == "AttnDownBlock2D":
        assert downsample_type == "conv" or not add_downsample
        return _Down(num_layers, in_channels, out_channels, add_downsample, resnet_eps, resnet_groups, attention_head_dim, downsample_padding)
    raise ValueError(f"{down_block_type} is not part of the stub")


def get_up_block(up_block_type, num_layers, in_channels, out_channels, prev_output_channel, temb_channels, add_upsample, resnet_eps,
                 resnet_act_fn, resnet_groups=None, attention_head_dim=None, resnet_time_scale_shift="default", upsample_type=None,
                 dropout=0.0, **unused):
    assert temb_channels is None and resnet_act_fn == "silu" and resnet_time_scale_shift == "default" and dropout == 0.0
    if up_block_type == "UpBlock2D":
        return _Up(num_layers, in_channels, out_channels, prev_output_channel, add_upsample, resnet_eps, resnet_groups, None)
    if up_block_type == "AttnUpBlock2D":
        assert upsample_type == "conv" or not add_upsample
        return _Up(num_layers, in_channels, out_channels, prev_output_channel, add_upsample, resnet_eps, resnet_groups, attention_head_dim)
    raise ValueError(f"{up_block_type} is not part of the stub")

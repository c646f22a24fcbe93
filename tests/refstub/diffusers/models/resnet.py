"""ResnetBlock2D / TemporalConvLayer / Downsample2D / Upsample2D -> oracle.layers, behind the diffusers-0.24 ctor
signatures the reference calls (models/unet_3d_blocks.py:262-328,425-481,564-596,660-712,794-822)."""
from oracle import layers as L


class ResnetBlock2D(L.ResnetBlock2D):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512, groups=32,
                 groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish", skip_time_act=False,
                 time_embedding_norm="default", kernel=None, output_scale_factor=1.0, use_in_shortcut=None, up=False,
                 down=False, conv_shortcut_bias=True, conv_2d_out_channels=None):
        assert time_embedding_norm == "default" and non_linearity in ("swish", "silu") and pre_norm and not up and not down
        assert dropout == 0.0 and groups_out is None and not conv_shortcut and not skip_time_act
        super().__init__(in_channels, out_channels, temb_channels, eps, groups, output_scale_factor)


class TemporalConvLayer(L.TemporalConvLayer):
    def __init__(self, in_dim, out_dim=None, dropout=0.0):
        super().__init__(in_dim, out_dim, dropout)


class Downsample2D(L.Downsample2D):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        assert use_conv and name == "op"
        super().__init__(channels, out_channels, padding)

    def forward(self, hidden_states, scale=1.0):
        return super().forward(hidden_states)


class Upsample2D(L.Upsample2D):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv"):
        assert use_conv and not use_conv_transpose
        super().__init__(channels, out_channels)

    def forward(self, hidden_states, output_size=None, scale=1.0):
        return super().forward(hidden_states, output_size)

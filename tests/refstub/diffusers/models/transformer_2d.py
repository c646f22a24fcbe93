"""Transformer2DModel -> oracle.layers (reference ctor sites: models/unet_3d_blocks.py:287,446,681)."""
from oracle import layers as L


class Transformer2DModel(L.Transformer2DModel):
    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, out_channels=None, num_layers=1,
                 dropout=0.0, norm_num_groups=32, cross_attention_dim=None, attention_bias=False, use_linear_projection=False,
                 only_cross_attention=False, upcast_attention=False, **kw):
        assert num_layers == 1 and use_linear_projection and not only_cross_attention and not upcast_attention and not kw
        super().__init__(num_attention_heads, attention_head_dim, in_channels, cross_attention_dim, norm_num_groups)

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, class_labels=None,
                cross_attention_kwargs=None, attention_mask=None, return_dict=True):
        assert timestep is None and class_labels is None and attention_mask is None
        return super().forward(hidden_states, encoder_hidden_states)

"""TransformerTemporalModel -> oracle.layers (reference ctor sites: models/unet_3d_condition_mask.py:163-168,
models/unet_3d_blocks.py:299,459,694)."""
from oracle import layers as L


class TransformerTemporalModel(L.TransformerTemporalModel):
    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, out_channels=None, num_layers=1,
                 dropout=0.0, norm_num_groups=32, cross_attention_dim=None, attention_bias=False, sample_size=None,
                 activation_fn="geglu", norm_elementwise_affine=True, double_self_attention=True):
        assert num_layers == 1 and activation_fn == "geglu" and double_self_attention and not attention_bias
        # (cross_attention_dim is accepted and - with double_self_attention - unused: both attentions are self-attentions)
        super().__init__(num_attention_heads, attention_head_dim, in_channels, norm_num_groups)

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, class_labels=None, num_frames=1,
                cross_attention_kwargs=None, return_dict=True):
        return super().forward(hidden_states, num_frames)

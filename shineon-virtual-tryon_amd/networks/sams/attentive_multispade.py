"""AttentiveMultiSpade: the SPADEs run in parallel on the same input, their outputs are stacked on the channel axis,
attended (SAGAN) and reduced back with a 3x3 conv + LeakyReLU() (reference: models/networks/sams/attentive_multispade.py)."""
from torch import nn

from ... import ops
from ..attention import ATTENTION_TYPES
from ..layers import HipConv2d
from .multispade import MultiSpade
from .spade import SPADE


class AttentiveMultiSpade(MultiSpade):
    def __init__(self, config_text, norm_nc, label_channels_dict, activation, attn_type="sagan"):
        super().__init__(config_text, norm_nc, label_channels_dict, activation)
        _, kernel_size = SPADE.parse_config_text(config_text)
        self.attn_nc = norm_nc * len(self.spade_layers)
        self.attention_layer = ATTENTION_TYPES[attn_type](self.attn_nc)
        # nn.Sequential(conv, nn.LeakyReLU()) in the reference: default slope 0.01, fused into the conv epilogue here
        self.mlp_final = nn.ModuleList([HipConv2d(self.attn_nc, norm_nc, kernel_size, padding=kernel_size // 2, fuse_leaky=0.01)])

    def forward(self, x, labelmap_dict, then_act=None):
        items = self.sort_fn(self._as_dict(labelmap_dict).items())
        stacked = ops.cat_channels([self.spade_layers[key](x, segmap) for key, segmap in items])
        y = self.mlp_final[0](self.attention_layer(stacked))
        return ops.activation(y, *then_act) if then_act is not None else y

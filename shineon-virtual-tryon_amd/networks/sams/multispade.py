"""MultiSpade: one SPADE per label map, applied one after another (reference: models/networks/sams/multispade.py)."""
from torch import Tensor, nn

from .spade import SPADE


class MultiSpade(SPADE):
    DEFAULT_KEY = "default_key"

    def __init__(self, config_text, norm_nc, label_channels_dict, activation, sort_fn=sorted):
        nn.Module.__init__(self)  # duck-types SPADE; none of its layers are wanted here
        if isinstance(label_channels_dict, int):
            label_channels_dict = {MultiSpade.DEFAULT_KEY: label_channels_dict}
        self.sort_fn = sort_fn
        self.label_channels = label_channels_dict
        self.spade_layers = nn.ModuleDict({key: SPADE(config_text, norm_nc, nc, activation)
                                           for key, nc in label_channels_dict.items()})

    def _as_dict(self, labelmaps):
        if isinstance(labelmaps, Tensor):
            if len(self.spade_layers) != 1:
                raise ValueError("You passed a single Tensor, but I don't know which spade layer to pass it through. "
                                 f"My spade layers are:\n{self.spade_layers}.")
            labelmaps = {next(iter(self.spade_layers.keys())): labelmaps}
        assert len(labelmaps) == len(self.spade_layers), f"{len(labelmaps)} != {len(self.spade_layers)}"
        return labelmaps

    def forward(self, x, labelmap_dict, then_act=None):
        items = self.sort_fn(self._as_dict(labelmap_dict).items())
        for n, (key, segmap) in enumerate(items):
            # only the LAST SPADE of the chain is followed by the residual block's activation
            x = self.spade_layers[key](x, segmap, then_act=then_act if n == len(items) - 1 else None)
        return x

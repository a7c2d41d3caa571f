"""Spectral normalisation of a HipConv2d (what `torch.nn.utils.spectral_norm(conv)` does to an nn.Conv2d in the
reference: models/networks/sams/spade.py:149-153, models/networks/normalization.py:24-25).

Checkpoint layout kept: the parameter is renamed `weight_orig`, the power-iteration vectors are the buffers
`weight_u` (O) and `weight_v` (I*R*S, torch's OIHW flattening), and — because the rename re-registers the parameter —
`bias` comes BEFORE `weight_orig` in `parameters()` / `state_dict()`.  Every training-mode forward performs one power
iteration (also under torch.no_grad()), eval-mode forwards use the stored vectors.
"""
import torch
from torch.nn import functional as F

from .. import ops, ops_sams
from .layers import HipConv2d


class SpectralHipConv2d(HipConv2d):
    """HipConv2d whose effective weight is weight_orig / sigma(weight_orig)."""

    def forward(self, x):
        w = ops_sams.spectral_normalize(self.weight_orig, self.weight_u, self.weight_v, self.training)
        return ops.conv2d(x, w, self.bias, self.stride, self.padding, self.fused_act, act_param=self.fused_act_param,
                          zero_bias_grad=self.zero_bias_grad == "always")

    def reset_parameters(self):
        if "weight_orig" in self._parameters:  # BaseNetwork.init_weights("none") on an already wrapped conv
            self._parameters["weight"] = self._parameters.pop("weight_orig")
            super().reset_parameters()
            self._parameters["weight_orig"] = self._parameters.pop("weight")
        else:
            super().reset_parameters()


def spectral_norm(conv):
    """In place: turn a freshly built HipConv2d into a SpectralHipConv2d (u, v ~ normalised N(0, 1) draws, as
    torch.nn.utils.spectral_norm initialises them)."""
    if not isinstance(conv, HipConv2d):
        raise TypeError("spectral_norm expects a HipConv2d")
    weight = conv._parameters.pop("weight")
    conv.register_parameter("weight_orig", weight)
    o = weight.shape[0]
    k = weight.numel() // o
    conv.register_buffer("weight_u", F.normalize(torch.empty(o).normal_(0, 1), dim=0, eps=1e-12))
    conv.register_buffer("weight_v", F.normalize(torch.empty(k).normal_(0, 1), dim=0, eps=1e-12))
    conv.__class__ = SpectralHipConv2d
    return conv

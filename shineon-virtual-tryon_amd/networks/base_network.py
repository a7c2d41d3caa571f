"""`BaseNetwork` of the SAMS-GAN networks (reference: models/networks/base_network.py).

It carries two things the reference's `train.py` relies on: the `--init_type / --init_variance` options and
`init_weights(init_type, gain)`, whose visiting order decides which value of the global torch generator lands in
which tensor.  The behaviour kept here (base_network.py:42-77):

  * every module is visited children-first (`nn.Module.apply`); a module whose class name contains "BatchNorm2d" gets
    scale ~ N(1, gain), shift 0 (only if it is affine); one whose name contains "Conv" or "Linear" and which has a
    `weight` gets the draw selected by `init_type` and a ZERO bias;
  * a spectrally normalised conv is initialised through `weight_orig` — in the reference `module.weight` still aliases
    that parameter's storage when `init_weights` runs (torch.nn.utils.spectral_norm sets it to `weight.data`);
  * after the pass, every direct child that has its own `init_weights` runs it AGAIN (so the sub-discriminators of the
    multiscale discriminator are drawn twice; the second draw is the one that stays).
"""
import math

import torch
from torch import nn

INIT_TYPES = ("normal", "xavier", "xavier_uniform", "kaiming", "orthogonal", "none")


def _fans(shape):
    receptive = 1
    for s in shape[2:]:
        receptive *= s
    return shape[1] * receptive, shape[0] * receptive


def _fill_logical(param, fill):
    """Run `fill` on a contiguous tensor of the parameter's logical shape and copy it over: conv weights live in OHWI
    memory, and torch's generators walk a permuted tensor differently from the contiguous one the reference owns."""
    buf = torch.empty(param.shape, dtype=param.dtype, device=param.device)
    fill(buf)
    param.copy_(buf)


def _main_weight(module):
    w = getattr(module, "weight_orig", None)
    return w if w is not None else getattr(module, "weight", None)


class BaseNetwork(nn.Module):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--init_type", type=str, default="xavier",
                            help="network initialization [normal|xavier|kaiming|orthogonal]")
        parser.add_argument("--init_variance", type=float, default=0.02, help="variance of the initialization distribution")
        return parser

    def print_network(self):
        millions = sum(p.numel() for p in self.parameters()) / 1e6
        print(f"Network [{type(self).__name__}] was created. Total number of parameters: {millions:.1f} million. "
              "To see the architecture, do print(network).")

    @torch.no_grad()
    def init_weights(self, init_type="normal", gain=0.02):
        def visit(m):
            name = type(m).__name__
            if "BatchNorm2d" in name:
                if getattr(m, "weight", None) is not None:
                    _fill_logical(m.weight, lambda t: t.normal_(1.0, gain))
                if getattr(m, "bias", None) is not None:
                    m.bias.fill_(0.0)
                return
            w = _main_weight(m)
            if w is None or not ("Conv" in name or "Linear" in name):
                return
            fan_in, fan_out = _fans(w.shape)
            if init_type == "normal":
                _fill_logical(w, lambda t: t.normal_(0.0, gain))
            elif init_type == "xavier":
                _fill_logical(w, lambda t: t.normal_(0.0, gain * math.sqrt(2.0 / (fan_in + fan_out))))
            elif init_type == "xavier_uniform":
                bound = math.sqrt(3.0) * math.sqrt(2.0 / (fan_in + fan_out))
                _fill_logical(w, lambda t: t.uniform_(-bound, bound))
            elif init_type == "kaiming":
                _fill_logical(w, lambda t: t.normal_(0.0, math.sqrt(2.0 / fan_in)))
            elif init_type == "orthogonal":
                _fill_logical(w, lambda t: nn.init.orthogonal_(t, gain=gain))
            elif init_type == "none":
                m.reset_parameters()
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if getattr(m, "bias", None) is not None:
                m.bias.fill_(0.0)

        self.apply(visit)
        for child in self.children():
            if hasattr(child, "init_weights"):
                child.init_weights(init_type, gain)

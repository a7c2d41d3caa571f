"""Normalisation wrapper of the PatchGAN discriminator layers (reference: models/networks/normalization.py:14-50).

`norm_type` is "spectral" + one of "" / "none" / "instance" / "batch" / "sync_batch".  The reference binds
`subnorm_type` only inside its `startswith("spectral")` branch, so any other value dies with UnboundLocalError at the
first wrapped layer; the same exception is raised here.
"""
from torch import nn

from .layers import HipBatchNorm2d, HipInstanceNorm2d
from .spectral import spectral_norm
from .sync_batchnorm import SynchronizedBatchNorm2d


def get_nonspade_norm_layer(opt, norm_type="instance"):
    def add_norm_layer(conv):
        if not norm_type.startswith("spectral"):
            raise UnboundLocalError("local variable 'subnorm_type' referenced before assignment")
        conv = spectral_norm(conv)
        subnorm = norm_type[len("spectral"):]
        if subnorm in ("none", ""):
            return conv
        # a bias in front of a normalisation has no effect: the reference deletes it (normalization.py:32-36)
        if conv.bias is not None:
            conv.register_parameter("bias", None)
        channels = conv.out_channels
        if subnorm == "batch":
            norm = HipBatchNorm2d(channels, affine=True)
        elif subnorm == "sync_batch":
            norm = SynchronizedBatchNorm2d(channels, affine=True)
        elif subnorm == "instance":
            norm = HipInstanceNorm2d(channels)
        else:
            raise ValueError("normalization layer %s is not recognized" % subnorm)
        return nn.Sequential(conv, norm)

    return add_norm_layer

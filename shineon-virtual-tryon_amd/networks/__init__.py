"""Weight initialisation of the hot-path networks (behaviour of the reference's `init_weights`,
models/networks/__init__.py:52-96, as used by UnetMaskModel.__init__ and FeatureExtraction.__init__).

What it must reproduce:
  * a module is classified by a fragment of its CLASS NAME, first match in the order Conv, Linear, BatchNorm2d — so
    SelfAttention's 1x1 convolutions are initialised too and InstanceNorm2d (no match) is left alone;
  * convolution / linear WEIGHTS are drawn from a zero-mean normal (std 0.02 for "normal"), their biases keep the
    constructor's default;  BatchNorm2d scale ~ N(1, 0.02), shift = 0;
  * modules are visited children-first (nn.Module.apply order) and every draw comes from the global torch generator,
    so with the same seed and the same module tree the values are the ones the reference produces
    (tests/test_abi_host_cpu.py::test_init_weights_matches_reference_stream).

The rule table below replaces the reference's three hand-written visitor functions.
"""
import math

import torch

# init_type -> standard deviation of the weight draw as a function of (fan_in, fan_out)
_WEIGHT_STD = {
    "normal": lambda fan_in, fan_out: 0.02,
    "xavier": lambda fan_in, fan_out: 0.02 * math.sqrt(2.0 / (fan_in + fan_out)),   # xavier_normal_, gain 0.02
    "kaiming": lambda fan_in, fan_out: math.sqrt(2.0 / fan_in),                      # kaiming_normal_, a=0, fan_in
}
_CONTRACTIONS = ("Conv", "Linear")  # class-name fragments whose `weight` gets the zero-mean draw
_BATCHNORM = "BatchNorm2d"


def _fans(weight):
    receptive = 1
    for s in weight.shape[2:]:
        receptive *= s
    return weight.shape[1] * receptive, weight.shape[0] * receptive


def _classify(module):
    name = type(module).__name__
    for fragment in _CONTRACTIONS:
        if fragment in name:
            return "contraction"
    return "batchnorm" if _BATCHNORM in name else None


def _draw(param, mean, std):
    """param <- N(mean, std) drawn in the parameter's LOGICAL (O, I, R, S) order.  Our conv weights live in OHWI memory
    (a permuted view); torch fills a non-contiguous tensor through a different generator path than a contiguous one,
    so the draw goes into a contiguous buffer of the logical shape - the tensor the reference's nn.Conv2d owns - and
    is copied over."""
    param.copy_(torch.empty(param.shape, dtype=param.dtype, device=param.device).normal_(mean, std))


def _children_first(module):
    for child in module.children():
        yield from _children_first(child)
    yield module


@torch.no_grad()
def init_weights(net, init_type="normal"):
    if init_type not in _WEIGHT_STD:
        raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
    std_of = _WEIGHT_STD[init_type]
    for m in _children_first(net):
        kind = _classify(m)
        if kind == "contraction":
            _draw(m.weight, 0.0, std_of(*_fans(m.weight)))
        elif kind == "batchnorm":
            _draw(m.weight, 1.0, 0.02)
            m.bias.fill_(0.0)
    return net

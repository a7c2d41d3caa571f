"""VGG19 feature slices for the perceptual loss (reference: models/networks/vgg.py:6-36).

Topology = torchvision vgg19().features[0:30] cut after relu1_1, 2_1, 3_1, 4_1, 5_1; module indices are
kept so `criterionVGG.vgg.sliceK.I.{weight,bias}` keys match.  Conv3x3 + ReLU is one MFMA kernel
(ReLU in the epilogue), the following ReLU slot is an identity placeholder.
The ImageNet weights need torchvision + a download; neither exists offline, so the weights are the
(frozen) default initialisation unless a state_dict is loaded on top.
"""
import logging

from torch import nn

from .layers import FusedReLU, HipConv2d, HipMaxPool2x2

logger = logging.getLogger("logger")

# torchvision cfg "E" up to conv5_1/relu5_1 (features[0:30])
_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512]


def _vgg19_features():
    layers, cin = [], 3
    for v in _CFG:
        if v == "M":
            layers.append(HipMaxPool2x2())
        else:
            layers += [HipConv2d(cin, v, kernel_size=3, stride=1, padding=1, fuse_relu=True), FusedReLU()]
            cin = v
    return layers


def _copy_into(layers, weights_by_index, what):
    """weights_by_index: {features index: (weight, bias)} in torchvision's vgg19().features numbering."""
    import torch

    convs = [(i, layer) for i, layer in enumerate(layers) if isinstance(layer, HipConv2d)]
    missing = [i for i, _ in convs if i not in weights_by_index]
    if missing:
        raise KeyError(f"{what}: no weights for VGG19 features layers {missing}")
    with torch.no_grad():
        for i, layer in convs:
            w, b = weights_by_index[i]
            if tuple(w.shape) != tuple(layer.weight.shape):
                raise ValueError(f"{what}: features.{i}.weight has shape {tuple(w.shape)}, expected {tuple(layer.weight.shape)}")
            layer.weight.copy_(w)
            layer.bias.copy_(b)


def load_vgg19_file(layers, path):
    """VGG19 weights from a file: either torchvision's vgg19 state_dict layout (`features.<i>.weight/bias`, what
    `torchvision.models.vgg19(pretrained=True).state_dict()` saves) or any checkpoint / state_dict that carries this
    model's own `...vgg.slice<k>.<i>.weight/bias` keys (a reference or shineon checkpoint)."""
    import re

    import torch

    sd = torch.load(path, map_location="cpu", weights_only=False)
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
    found = {}
    for k, v in sd.items():
        m = re.search(r"(?:^|\.)features\.(\d+)\.(weight|bias)$", k) or re.search(r"vgg\.slice\d\.(\d+)\.(weight|bias)$", k)
        if m:
            found.setdefault(int(m.group(1)), {})[m.group(2)] = v
    _copy_into(layers, {i: (d["weight"], d["bias"]) for i, d in found.items() if "weight" in d and "bias" in d}, path)


def _try_load_pretrained(layers, weights_file=None):
    """ImageNet weights: an explicit file (--vgg_weights) wins; else torchvision's download cache if torchvision is
    installed (models/networks/vgg.py:9 does vgg19(pretrained=True)).  Returns True when real weights were loaded."""
    if weights_file:
        load_vgg19_file(layers, weights_file)  # errors propagate: an explicit request must not fall back silently
        return True
    try:
        from torchvision import models  # not installed on the build image

        feats = models.vgg19(pretrained=True).features
        _copy_into(layers, {i: (m.weight.data, m.bias.data) for i, m in enumerate(feats) if hasattr(m, "weight")},
                   "torchvision vgg19")
        return True
    except Exception as e:  # noqa: BLE001
        logger.warning("VGG19 ImageNet weights unavailable (%s); the perceptual loss holds frozen RANDOM weights until "
                       "--vgg_weights <file> is given (training refuses to start without --allow_random_vgg)",
                       type(e).__name__)
        return False


class Vgg19(nn.Module):
    def __init__(self, requires_grad=False, pretrained=True, weights_file=None):
        super().__init__()
        feats = _vgg19_features()
        self.pretrained_loaded = _try_load_pretrained(feats, weights_file) if (pretrained or weights_file) else False
        self.slice1, self.slice2, self.slice3 = nn.Sequential(), nn.Sequential(), nn.Sequential()
        self.slice4, self.slice5 = nn.Sequential(), nn.Sequential()
        for lo, hi, sl in ((0, 2, self.slice1), (2, 7, self.slice2), (7, 12, self.slice3), (12, 21, self.slice4),
                           (21, 30, self.slice5)):
            for x in range(lo, hi):
                sl.add_module(str(x), feats[x])
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, X):
        h1 = self.slice1(X)
        h2 = self.slice2(h1)
        h3 = self.slice3(h2)
        h4 = self.slice4(h3)
        h5 = self.slice5(h4)
        return [h1, h2, h3, h4, h5]

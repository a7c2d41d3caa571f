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


def _try_load_pretrained(layers):
    try:
        from torchvision import models  # not installed on the build image

        feats = models.vgg19(pretrained=True).features
        for i, layer in enumerate(layers):
            if isinstance(layer, HipConv2d):
                layer.weight.data.copy_(feats[i].weight.data)
                layer.bias.data.copy_(feats[i].bias.data)
        return True
    except Exception as e:  # noqa: BLE001
        logger.warning("VGG19 ImageNet weights unavailable (%s); using frozen random weights", type(e).__name__)
        return False


class Vgg19(nn.Module):
    def __init__(self, requires_grad=False, pretrained=True):
        super().__init__()
        feats = _vgg19_features()
        self.pretrained_loaded = _try_load_pretrained(feats) if pretrained else False
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

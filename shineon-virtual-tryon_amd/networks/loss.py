"""Perceptual loss (reference: models/networks/loss.py:106-122): sum_i w_i * mean|vgg_i(x) - vgg_i(y)|
with w = (1/32, 1/16, 1/8, 1/4, 1), target branch detached.  GANLoss is SAMS-only and out of scope."""
import torch
from torch import nn

from .. import ops
from .vgg import Vgg19


class VGGLoss(nn.Module):
    def __init__(self, layids=None, pretrained=True):
        super().__init__()
        self.vgg = Vgg19(pretrained=pretrained)
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]
        self.layids = layids

    def forward(self, x, y):
        x_vgg = self.vgg(x)
        with torch.no_grad():
            y_vgg = self.vgg(y.detach())
        if self.layids is None:
            self.layids = list(range(len(x_vgg)))
        loss = 0
        for i in self.layids:
            loss = loss + ops.l1_loss(x_vgg[i], y_vgg[i], self.weights[i])
        return loss

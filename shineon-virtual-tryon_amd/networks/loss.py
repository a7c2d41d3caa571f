"""Perceptual loss (reference: models/networks/loss.py:106-122): sum_i w_i * mean|vgg_i(x) - vgg_i(y)|
with w = (1/32, 1/16, 1/8, 1/4, 1), target branch detached; and the SAMS-GAN's adversarial loss (loss.py:13-103)."""
import torch
from torch import nn

from .. import ops, ops_sams
from .vgg import Vgg19


class GANLoss(nn.Module):
    """LSGAN / cross-entropy / Wasserstein / hinge objective on discriminator predictions (loss.py:13-103) with the
    default labels (real 1, fake 0).  A list of predictions is averaged over its entries — an entry that is itself a
    list (one discriminator's stage outputs) contributes its LAST element — and the list branch returns shape (1,),
    like the reference."""

    AVAILABLE_MODES = ("ls", "original", "w", "hinge")

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0, tensor=None, opt=None):
        super().__init__()
        assert gan_mode in GANLoss.AVAILABLE_MODES, f"Unexpected gan_mode = {gan_mode}"
        if (target_real_label, target_fake_label) != (1.0, 0.0):
            raise NotImplementedError("GANLoss labels other than 1.0 / 0.0 (the reference never passes any)")
        self.gan_mode, self.opt = gan_mode, opt

    def loss(self, input, target_is_real, for_discriminator=True):
        return ops_sams.gan_loss(input, self.gan_mode, target_is_real, for_discriminator)

    def __call__(self, input, target_is_real, for_discriminator=True):
        if isinstance(input, list):
            total = 0
            for pred in input:
                if isinstance(pred, list):
                    pred = pred[-1]
                total = total + self.loss(pred, target_is_real, for_discriminator).reshape(1)
            return total / len(input)
        return self.loss(input, target_is_real, for_discriminator)


class VGGLoss(nn.Module):
    def __init__(self, layids=None, pretrained=True, weights_file=None):
        super().__init__()
        self.vgg = Vgg19(pretrained=pretrained, weights_file=weights_file)
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]
        self.layids = layids

    def _fused_plan(self):
        """(cfg, params) for ops.vgg_perceptual_loss: the VGG stack flattened in execution order with the tap
        weight attached to the conv whose ReLU output is compared (relu1_1, 2_1, 3_1, 4_1, 5_1)."""
        cfg, params = [], []
        for k, sl in enumerate((self.vgg.slice1, self.vgg.slice2, self.vgg.slice3, self.vgg.slice4, self.vgg.slice5)):
            convs = [m for m in sl if hasattr(m, "weight")]
            for m in sl:
                if hasattr(m, "weight"):
                    is_tap = m is convs[-1] and (self.layids is None or k in self.layids)
                    cfg.append(("C", self.weights[k] if is_tap else None))
                    params += [m.weight, m.bias]
                elif m.__class__.__name__ == "HipMaxPool2x2":
                    cfg.append(("M",))
        return tuple(cfg), params

    def _fusable(self, t):
        return t.is_cuda and t.shape[1] == 3 and not any(p.requires_grad for p in self.vgg.parameters())

    def target_features(self, y):
        """Tap features of the target image, to be handed to forward(x, y, y_features=...): they depend on the batch only,
        so a training pipeline can compute them ahead of the step on another stream."""
        if not self._fusable(y):
            raise RuntimeError("target_features needs a 3-channel CUDA image and frozen VGG weights")
        cfg, params = self._fused_plan()
        return ops.vgg_target_features(y, cfg, params)

    def forward(self, x, y, y_features=None):
        if self._fusable(x):
            cfg, params = self._fused_plan()
            return ops.vgg_perceptual_loss(x, y, cfg, params, y_features=y_features)
        x_vgg = self.vgg(x)
        with torch.no_grad():
            y_vgg = self.vgg(y.detach())
        if self.layids is None:
            self.layids = list(range(len(x_vgg)))
        loss = 0
        for i in self.layids:
            loss = loss + ops.l1_loss(x_vgg[i], y_vgg[i], self.weights[i])
        return loss

"""Minimal stand-ins for the pytorch-lightning 0.9 objects the reference's models touch
(reference: models/base_model.py:24, models/warp_model.py:9,94-97, models/unet_mask_model.py:8,196-215).
pytorch_lightning is not a dependency of this package; the Trainer in trainer.py drives the same hooks.
"""
import argparse

import torch
from torch import nn


class _Result:
    def __init__(self):
        self.logs = {}
        self.prog_bar = {}

    def log(self, name, value, prog_bar=False, **_):
        self.logs[name] = value
        if prog_bar:
            self.prog_bar[name] = value


class TrainResult(_Result):
    def __init__(self, minimize=None, **_):
        super().__init__()
        self.minimize = minimize


class EvalResult(_Result):
    def __init__(self, checkpoint_on=None, early_stop_on=None, **_):
        super().__init__()
        self.checkpoint_on = checkpoint_on
        self.early_stop_on = early_stop_on


class _NullExperiment:
    def add_image(self, *a, **k):
        pass

    def add_text(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass


class NullLogger:
    experiment = _NullExperiment()


class LightningModule(nn.Module):
    """The slice of pl.LightningModule the hot path relies on."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.global_step = 0
        self.current_epoch = 0
        self.logger = NullLogger()
        self.trainer = None

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location="cpu", **kwargs):
        """Lightning-format checkpoint: {'state_dict': ..., 'hparams' | 'hyper_parameters': ...}."""
        ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
        hp = ckpt.get("hyper_parameters", ckpt.get("hparams"))
        if hp is None:
            raise KeyError("checkpoint has no hparams / hyper_parameters entry")
        if isinstance(hp, dict):
            hp = argparse.Namespace(**hp)
        for k, v in kwargs.items():
            setattr(hp, k, v)
        model = cls(hp)
        model.load_state_dict(ckpt["state_dict"], strict=False)
        return model

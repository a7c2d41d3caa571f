"""Shared model surface (reference: models/base_model.py:24-237): CLI flags, channel bookkeeping,
hparams override, dataloaders with a DistributedSampler, Adam + linear-decay LambdaLR, validation glue.
Lightning is replaced by pl_compat.LightningModule + trainer.Trainer, which call the same hooks.
"""
import abc
import argparse
import logging
import os.path as osp

import torch
from torch.utils.data import DataLoader

from .optim import HipAdam
from .pl_compat import LightningModule
from .tryon_channels import RGB_CHANNELS, parse_num_channels

logger = logging.getLogger("logger")


class BaseModel(LightningModule, abc.ABC):
    def load_state_dict(self, state_dict, strict=True, **kw):
        """nn.Module.load_state_dict, then the derived-weight caches of ops.py are dropped (a load that goes through `.data`
        leaves `_version` alone; the caches would keep serving transforms of the old weights to gradient-free passes)."""
        from . import ops

        out = super().load_state_dict(state_dict, strict=strict, **kw)
        ops.invalidate_weight_caches()
        return out

    @classmethod
    def modify_commandline_options(cls, parser: argparse.ArgumentParser, is_train):
        parser.add_argument("--person_inputs", nargs="+",
                            help="List of what type of items are passed as person input.")
        parser.add_argument("--cloth_inputs", nargs="+", default=("cloth",),
                            help="List of items to pass as the cloth inputs.")
        parser.add_argument("--ngf", type=int, default=64)
        parser.add_argument("--self_attn", action="store_true", help="Add self-attention")
        parser.add_argument("--no_self_attn", action="store_false", dest="self_attn", help="No self-attention")
        parser.add_argument("--num_attn", type=int, default=2,
                            help="Num of self-attention layers, from the bottom of the UNet upwards")
        parser.add_argument("--flow_warp", action="store_true", help="Warp the previous frame with flow")
        return parser

    def __init__(self, hparams, *args, **kwargs):
        if isinstance(hparams, dict):
            hparams = argparse.Namespace(**hparams)
        super().__init__(*args, **kwargs)
        self.hparams = hparams
        self.n_frames_total = hparams.n_frames_total
        self.person_channels = parse_num_channels(hparams.person_inputs)
        self.cloth_channels = parse_num_channels(hparams.cloth_inputs)
        self.is_train = self.hparams.is_train
        if self.is_train:
            self.val_visualization_batch = None

    def override_hparams(self, hparams):
        """Re-apply non-architectural flags after a checkpoint load (base_model.py:76-89).
        One deliberate difference: `is_train` follows the NEW options.  The reference keeps the value stored in the checkpoint
        (True for anything train.py wrote), so `python test.py --checkpoint <train checkpoint>` never sets test_results_dir and
        its test_step dies on the missing attribute (the TODO at train.py:41-44); here the documented test command lines run."""
        self.hparams = hparams
        self.is_train = bool(getattr(hparams, "is_train", self.is_train))
        if not self.is_train:
            ckpt_name = osp.basename(hparams.checkpoint)
            self.test_results_dir = osp.join(hparams.result_dir, hparams.name, ckpt_name, hparams.datamode)

    # ---- data ------------------------------------------------------------------------------------
    def prepare_data(self):
        pass

    def setup(self, stage):
        from .data import find_dataset_using_name

        dataset_cls = find_dataset_using_name(self.hparams.dataset)
        self.train_dataset = dataset_cls(self.hparams)
        logger.info(f"Main {self.hparams.dataset} dataset initialized: {len(self.train_dataset)} samples.")
        if stage == "fit":
            self.val_dataset = self.train_dataset.make_validation_dataset(self.hparams)

    def _sampler(self, dataset):
        shuffle = not getattr(self.hparams, "no_shuffle", False)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.utils.data.distributed.DistributedSampler(dataset, shuffle=shuffle)
        return torch.utils.data.RandomSampler(dataset) if shuffle else torch.utils.data.SequentialSampler(dataset)

    def train_dataloader(self):
        return DataLoader(self.train_dataset, batch_size=self.hparams.batch_size,
                          sampler=self._sampler(self.train_dataset), num_workers=self.hparams.workers)

    def val_dataloader(self):
        return DataLoader(self.val_dataset, batch_size=self.hparams.batch_size,
                          sampler=self._sampler(self.val_dataset), num_workers=self.hparams.workers)

    def test_dataloader(self):
        return DataLoader(self.train_dataset, batch_size=self.hparams.batch_size, num_workers=self.hparams.workers)

    # ---- steps -----------------------------------------------------------------------------------
    def validation_step(self, batch, idx):
        self.val_visualization_batch = batch
        return self.training_step(batch, idx, val=True)

    def visualize(self, input_batch, tag="train"):
        """Outputs to visualise are stashed on self by training_step; TensorBoard itself is out of scope."""

    def on_validation_epoch_end(self):
        if self.val_visualization_batch is not None:
            self.visualize(self.val_visualization_batch, "validation")

    # ---- optimisation ----------------------------------------------------------------------------
    def configure_optimizers(self):
        # modules may ask for parameters to sit next to each other in the optimizer's flat slab (SelfAttention: q/k/v)
        adjacent = [g for m in self.modules() if hasattr(m, "adjacent_param_groups") for g in m.adjacent_param_groups()]
        optimizer = HipAdam([p for p in self.parameters() if p.requires_grad], self.hparams.lr, adjacent=adjacent)
        scheduler = self._make_step_scheduler(optimizer)
        return [optimizer], [scheduler]

    def _make_step_scheduler(self, optimizer):
        keep, decay_epochs = self.hparams.keep_epochs, self.hparams.decay_epochs

        def step_func(epoch):
            return 1.0 - max(0, epoch - keep) / float(decay_epochs + 1)

        return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=step_func)

    # ---- visual helpers (base_model.py:186-237) -----------------------------------------------------
    def replace_actual_with_visual(self):
        names = list(self.hparams.person_inputs)
        if "agnostic" in names:
            i = names.index("agnostic")
            names[i:i + 1] = ["silhouette", "im_head"]
        if "cocopose" in names:
            names[names.index("cocopose")] = "im_cocopose"
        if "flow" in names:
            i = names.index("flow")
            names.pop(i)
            if getattr(self.hparams, "visualize_flow", False):
                names.insert(i, "flow_image")
        return names

    def fetch_person_visuals(self, batch, sort_fn=None):
        names = self.replace_actual_with_visual()
        if sort_fn:
            names = sort_fn(names)
        out = [batch[n] for n in names if n in batch and batch[n].shape[-3] <= RGB_CHANNELS]
        if not out:
            raise ValueError("Didn't find any tensors to visualize!")
        return out

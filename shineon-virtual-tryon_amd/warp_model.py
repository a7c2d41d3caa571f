"""WarpModel — the geometric matching module (reference: models/warp_model.py:27-152).

forward: features(person), features(cloth) -> L2 norm -> correlation -> regression -> theta -> TPS grid.
training_step: L1(grid_sample(cloth, grid, border), im_cloth).
"""
import argparse
import os
from argparse import ArgumentParser

from torch import nn

from . import ops
from .base_model import BaseModel
from .io_png import StageWriter
from .networks.layers import HipBatchNorm2d, share_bn_counters
from .networks.cpvton.warp import (FeatureCorrelation, FeatureExtraction, FeatureL2Norm, FeatureRegression,
                                   TpsGridGen)
from .pl_compat import EvalResult, TrainResult
from .util import get_and_cat_inputs, maybe_combine_frames_and_channels


# the two towers on two streams (see WarpModel.forward); SHINEON_TOWER_STREAMS=0 runs them one after the other
TOWER_STREAMS = os.environ.get("SHINEON_TOWER_STREAMS", "1") != "0"


class WarpModel(BaseModel):
    """ Geometric Matching Module """

    @classmethod
    def modify_commandline_options(cls, parser: ArgumentParser, is_train):
        parser = ArgumentParser(parents=[parser], add_help=False)
        parser = super(WarpModel, cls).modify_commandline_options(parser, is_train)
        parser.add_argument("--grid_size", type=int, default=5)
        parser.set_defaults(person_inputs=("agnostic", "cocopose"))
        return parser

    def __init__(self, hparams):
        super().__init__(hparams)
        if isinstance(hparams, dict):
            hparams = argparse.Namespace(**hparams)
        self.extractionA = FeatureExtraction(self.person_channels, ngf=hparams.ngf, n_layers=3)
        self.extractionB = FeatureExtraction(self.cloth_channels, ngf=hparams.ngf, n_layers=3)
        self.l2norm = FeatureL2Norm()
        self.correlation = FeatureCorrelation()
        self.regression = FeatureRegression(input_nc=192, output_dim=2 * hparams.grid_size ** 2)
        self.gridGen = TpsGridGen(hparams.fine_height, hparams.fine_width, grid_size=hparams.grid_size)
        # the cloth tower on a forked stream (forward()).  Measured on MI355X, bs=4: WarpModel alone 1610 -> 1908 frames/s; inside
        # the chained warp -> try-on schedule (whose warp stage already shares the chip with the try-on stage on a second
        # stream) 612 -> 521, so trainer.ChainedTrainStep switches it off for its warp model.
        self.tower_streams = TOWER_STREAMS

    def batch_keys(self):
        """Tensor entries of the batch dict this model reads (training / validation / test)."""
        return set(self.hparams.person_inputs) | set(self.hparams.cloth_inputs) | {"cloth", "im_cloth", "cloth_mask", "grid_vis"}

    def plant_shared_buffers(self):
        """Re-home the 14 num_batches_tracked counters as views of ONE int64 tensor (layers.share_bn_counters) and
        return it; idempotent until the buffers are moved by .to()."""
        flat = getattr(self, "_bn_flat", None)
        first = next((m for m in self.modules() if isinstance(m, HipBatchNorm2d)), None)
        if first is None:
            return None
        if flat is None or first.num_batches_tracked.data_ptr() != flat.data_ptr():  # first use, or moved by .to()
            flat = share_bn_counters(self)
            object.__setattr__(self, "_bn_flat", flat)
        return flat

    def _bump_bn_counters(self):
        """num_batches_tracked += 1 for all 14 BatchNorm layers with one launch."""
        flat = self.plant_shared_buffers()
        if flat is not None:
            flat.add_(1)

    def forward(self, inputA, inputB):
        if self.training:
            self._bump_bn_counters()
        if self.tower_streams and inputA.is_cuda:
            # The two feature-extraction towers (warp.py:9-36) are independent until the correlation and, at 16x12 .. 128x96
            # feature maps, every one of their kernels fills a fraction of the 256 CUs: the cloth tower runs on a forked stream
            # with its own scratch lane (eagerly and as a parallel branch of a captured graph); autograd replays the same
            # two-stream structure in the backward pass (each node on the stream and scratch lane of its forward op).
            fork = ops._SideStream(inputA.device)
            with fork, ops.workspace_lane(ops._LANE_BASE[0] + 16):
                featureB = self.l2norm(self.extractionB(inputB))
            featureA = self.l2norm(self.extractionA(inputA), transpose_hw=True)  # the h<->w transpose of warp.py:60 is fused here
            fork.join()
        else:
            featureA = self.extractionA(inputA)
            featureB = self.extractionB(inputB)
            featureA = self.l2norm(featureA, transpose_hw=True)  # the h<->w transpose of warp.py:60 is fused here
            featureB = self.l2norm(featureB)
        correlation = self.correlation(featureA, featureB, a_is_transposed=True)
        theta = self.regression(correlation)
        grid = self.gridGen(theta)
        return grid, theta

    def training_step(self, batch, idx, val=False):
        batch = maybe_combine_frames_and_channels(self.hparams, batch)
        c, im_c = batch["cloth"], batch["im_cloth"]
        person_inputs = get_and_cat_inputs(batch, self.hparams.person_inputs)
        cloth_inputs = get_and_cat_inputs(batch, self.hparams.cloth_inputs)

        grid, theta = self.forward(person_inputs, cloth_inputs)
        warped_cloth = ops.grid_sample(c, grid, padding_mode="border")
        # stashed for visualisation / the next stage only: detached, so the model never pins an autograd graph
        self.warped_cloth = warped_cloth.detach()
        if "grid_vis" in batch:  # visual only, no loss (warp_model.py:86)
            self.warped_grid = ops.grid_sample(batch["grid_vis"], grid.detach(), padding_mode="zeros")
        loss = ops.l1_loss(warped_cloth, im_c)

        if not val and self.global_step % self.hparams.display_count == 0:
            self.visualize(batch)
        val_ = "val_" if val else ""
        result = EvalResult(checkpoint_on=loss) if val else TrainResult(loss)
        result.log(f"{val_}loss/G", loss, prog_bar=True)
        return result

    def test_step(self, batch, batch_idx):
        """Writes warp-cloth/ (and warp-mask/) images for the try-on stage (warp_model.py:115-152); skipped when the
        warped cloths of this batch already exist."""
        batch = maybe_combine_frames_and_channels(self.hparams, batch)
        writer = StageWriter(self.test_results_dir, batch["dataset_name"], batch["cloth_name"], primary="warp-cloth")

        def produce():
            grid, _ = self.forward(get_and_cat_inputs(batch, self.hparams.person_inputs),
                                   get_and_cat_inputs(batch, self.hparams.cloth_inputs))
            self.warped_cloth = ops.grid_sample(batch["cloth"], grid, padding_mode="border")
            warped_mask = ops.grid_sample(batch["cloth_mask"], grid, padding_mode="zeros")
            if "grid_vis" in batch:
                self.warped_grid = ops.grid_sample(batch["grid_vis"], grid, padding_mode="zeros")
            return {"warp-cloth": self.warped_cloth, "warp-mask": warped_mask * 2 - 1}

        return writer.run(produce)

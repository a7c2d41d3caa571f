"""UnetMaskModel — the CP-VTON try-on module (reference: models/unet_mask_model.py:27-326).

forward: cat(person, cloth) -> U-Net -> tanh (rendered person) | sigmoid (try-on mask) [| sigmoid (flow
mask)] -> p_tryon = (1 - mask) * p_rendered + mask * warped_cloth, per frame.
training_step: L1(p_tryon, image) + VGG(p_tryon, image) + L1(mask, cloth_mask) + pen * sum(flow_mask).
"""
import argparse
import logging
import math

import torch

from . import ops
from .base_model import BaseModel
from .io_png import StageWriter
from .networks import init_weights
from .networks.cpvton.unet import UnetGenerator
from .networks.layers import HipInstanceNorm2d
from .networks.loss import VGGLoss
from .pl_compat import EvalResult, TrainResult
from .tryon_channels import RGB_CHANNELS
from .util import get_and_cat_inputs, maybe_combine_frames_and_channels

logger = logging.getLogger("logger")


class Resample2d(torch.nn.Module):
    """Flow warp of the previous generated frame (flownet2 Resample2d, kernel_size=1, bilinear)."""

    def forward(self, input1, input2):
        return ops.resample2d(input1, input2)


class UnetMaskModel(BaseModel):
    """ CP-VTON Try-On Module (TOM) """

    @classmethod
    def modify_commandline_options(cls, parser: argparse.ArgumentParser, is_train):
        parser = argparse.ArgumentParser(parents=[parser], add_help=False)
        parser = super(UnetMaskModel, cls).modify_commandline_options(parser, is_train)
        parser.set_defaults(person_inputs=("agnostic", "densepose"))
        parser.add_argument("--pen_flow_mask", type=float, default=1.0, help="Penalty applied to flow mask loss")
        parser.add_argument("--vgg_weights", default=None, help="file with ImageNet VGG19 weights for the perceptual loss: "
                            "torchvision's vgg19 state_dict (features.N.weight/bias) or a checkpoint holding criterionVGG.*")
        parser.add_argument("--allow_random_vgg", action="store_true", help="train even though no pretrained VGG19 weights "
                            "could be loaded (synthetic benchmarks / tests only: the perceptual term is then meaningless)")
        return parser

    def __init__(self, hparams):
        super().__init__(hparams)
        if isinstance(hparams, dict):
            hparams = argparse.Namespace(**hparams)
        self.hparams = hparams
        n_frames = hparams.n_frames_total if hasattr(hparams, "n_frames_total") else 1
        self.unet = UnetGenerator(
            input_nc=(self.person_channels + self.cloth_channels) * n_frames,
            output_nc=5 * n_frames if self.hparams.flow_warp else 4 * n_frames,
            num_downs=6,
            num_attention=hparams.num_attn if hasattr(hparams, "num_attn") else 2,
            ngf=int(64 * (math.log(n_frames) + 1)),
            norm_layer=HipInstanceNorm2d,
            use_self_attn=hparams.self_attn,
            activation=hparams.activation,
        )
        self.resample = Resample2d()
        self.criterionVGG = VGGLoss(weights_file=getattr(hparams, "vgg_weights", None))
        init_weights(self.unet, init_type="normal")

    def batch_keys(self):
        """Tensor entries of the batch dict this model reads (training / validation / test)."""
        keys = set(self.hparams.person_inputs) | set(self.hparams.cloth_inputs) | {"image", "prev_image", "cloth_mask"}
        return keys | ({"flow"} if self.hparams.flow_warp else set())

    def require_pretrained_vgg(self):
        """Called by Trainer.fit before training starts: the reference always trains against ImageNet VGG19 features
        (models/networks/vgg.py:9); refuse to optimise against random ones unless explicitly allowed.  A checkpoint
        loaded afterwards brings its own criterionVGG.* weights (they are part of the state_dict)."""
        if not self.criterionVGG.vgg.pretrained_loaded and not getattr(self.hparams, "allow_random_vgg", False):
            raise RuntimeError("UnetMaskModel: no pretrained VGG19 weights for the perceptual loss (torchvision is not "
                               "installed / offline). Pass --vgg_weights <vgg19 state_dict or checkpoint>, or "
                               "--allow_random_vgg for synthetic benchmarks and tests.")

    def forward(self, person_representation, warped_cloths, flows=None, prev_im=None):
        n = self.hparams.n_frames_total
        concat_tensor = ops.cat_channels([person_representation, warped_cloths])
        outputs = self.unet(concat_tensor)

        if n == 1 and not self.hparams.flow_warp:
            # single frame: tanh | sigmoid | blend fused in one kernel
            p_rendereds, tryon_masks, p_tryons = ops.tryon_compose(outputs, warped_cloths)
            return p_rendereds, tryon_masks, p_tryons, None

        boundary, weight_boundary = 3 * n, 4 * n
        p_rendereds = ops.activation(outputs[:, 0:boundary], "tanh")
        tryon_masks = ops.activation(outputs[:, boundary:weight_boundary], "sigmoid")
        flow_masks = ops.activation(outputs[:, weight_boundary:], "sigmoid") if self.hparams.flow_warp else None

        flows = list(torch.chunk(flows, n, dim=1)) if flows is not None else None
        cloths = list(torch.chunk(warped_cloths, n, dim=1))
        rend = list(torch.chunk(p_rendereds, n, dim=1))
        masks = list(torch.chunk(tryon_masks, n, dim=1))
        fmasks = list(torch.chunk(flow_masks, n, dim=1)) if flow_masks is not None else None

        frames = []
        for f in range(n):
            if flows is not None and f > 0:
                warped_by_flow = self.resample(frames[f - 1], flows[f])
                p_rendered = ops.blend(warped_by_flow, rend[f], fmasks[f])
            else:
                p_rendered = rend[f]
            frames.append(ops.blend(p_rendered, cloths[f], masks[f]))
        p_tryons = ops.cat_channels(frames)
        return p_rendereds, tryon_masks, p_tryons, flow_masks

    def training_step(self, batch, batch_idx, val=False):
        hp = self.hparams
        n = hp.n_frames_total
        batch = maybe_combine_frames_and_channels(hp, batch)
        im, prev_im, cm = batch["image"], batch.get("prev_image"), batch["cloth_mask"]
        flow = batch["flow"] if hp.flow_warp else None
        person_inputs = get_and_cat_inputs(batch, hp.person_inputs)
        cloth_inputs = get_and_cat_inputs(batch, hp.cloth_inputs)

        p_rendereds, tryon_masks, p_tryons, flow_masks = self.forward(person_inputs, cloth_inputs, flow, prev_im)
        pt = torch.chunk(p_tryons, n, dim=1)
        tm = torch.chunk(tryon_masks, n, dim=1)
        fm = torch.chunk(flow_masks, n, dim=1) if flow_masks is not None else None
        # stashed for visualisation only: detached, so the model never pins an autograd graph
        self.p_tryons = tuple(t.detach() for t in pt)
        self.p_rendereds = tuple(t.detach() for t in torch.chunk(p_rendereds, n, dim=1))
        self.tryon_masks = tuple(t.detach() for t in tm)
        self.flow_masks = tuple(t.detach() for t in fm) if fm is not None else None
        im = torch.chunk(im, n, dim=1)
        cm = torch.chunk(cm, n, dim=1)

        def both(fn):
            """term on the last frame; averaged with the one before it when n > 1 (unet_mask_model.py:174-184)"""
            curr = fn(-1)
            if n > 1:
                prev = fn(-2)
                return 0.5 * (curr + prev), curr, prev
            return curr, curr, (ops.zero_scalar(curr.device) if curr.is_cuda else torch.zeros_like(curr))

        loss_image_l1, l1_curr, l1_prev = both(lambda i: ops.l1_loss(pt[i], im[i]))
        # a pipeline may have computed the target's VGG features ahead of the step (single-frame case only)
        yf = getattr(self, "vgg_target_features", None) if n == 1 else None
        loss_image_vgg, vgg_curr, vgg_prev = both(lambda i: self.criterionVGG(pt[i], im[i], y_features=yf))
        loss_tryon_mask_l1, m_curr, m_prev = both(lambda i: ops.l1_loss(tm[i], cm[i]))
        if fm is not None:
            loss_flow_mask_l1 = ops.tensor_sum(fm[-1]) * hp.pen_flow_mask
        elif m_curr.is_cuda:
            loss_flow_mask_l1 = ops.zero_scalar(m_curr.device)   # zeros_like(...) * pen_flow_mask (unet_mask_model.py:186-188)
        else:
            loss_flow_mask_l1 = torch.zeros_like(m_curr) * hp.pen_flow_mask

        loss = ops.scalar_sum(loss_image_l1, loss_image_vgg, loss_tryon_mask_l1, loss_flow_mask_l1)

        if not val and self.global_step % hp.display_count == 0:
            self.visualize(batch)
        val_ = "val_" if val else ""
        result = EvalResult(checkpoint_on=loss) if val else TrainResult(loss)
        result.log(f"{val_}loss/G", loss, prog_bar=True)
        result.log(f"{val_}loss/G/l1", loss_image_l1, prog_bar=True)
        result.log(f"{val_}loss/G/vgg", loss_image_vgg, prog_bar=True)
        result.log(f"{val_}loss/G/tryon_mask_l1", loss_tryon_mask_l1, prog_bar=True)
        result.log(f"{val_}loss/G/flow_mask_l1", loss_flow_mask_l1, prog_bar=True)
        if n > 1:
            result.log(f"{val_}loss/G/l1_prev", l1_prev)
            result.log(f"{val_}loss/G/vgg_prev", vgg_prev)
            result.log(f"{val_}loss/G/tryon_mask_prev", m_prev)
            result.log(f"{val_}loss/G/l1_curr", l1_curr)
            result.log(f"{val_}loss/G/vgg_curr", vgg_curr)
            result.log(f"{val_}loss/G/tryon_mask_curr", m_curr)
        return result

    def test_step(self, batch, batch_idx):
        """Writes the generated frames under tryon/ (with --tryon_list) or reconstruction/ (unet_mask_model.py:250-282);
        for n_frames_total > 1 the names of the LAST frame are used and only its RGB channels are saved."""
        hp = self.hparams
        batch = maybe_combine_frames_and_channels(hp, batch)
        last = (lambda seqs: [seq[-1] for seq in seqs]) if hp.n_frames_total > 1 else (lambda names: names)
        task = "tryon" if getattr(hp, "tryon_list", None) else "reconstruction"
        writer = StageWriter(self.test_results_dir, last(batch["dataset_name"]), last(batch["image_name"]), primary=task)

        def produce():
            _, _, self.p_tryon, _ = self.forward(get_and_cat_inputs(batch, hp.person_inputs),
                                                 get_and_cat_inputs(batch, hp.cloth_inputs))
            return {task: self.p_tryon[:, -RGB_CHANNELS:]}

        return writer.run(produce)

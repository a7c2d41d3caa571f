"""SamsModel — the Self-Attentive Multi-SPADE video try-on GAN (reference: models/sams_model.py).

Three optimizers, stepped in this order for every batch (Lightning's multi-optimizer loop; trainer.MultiOptimizerStep
reproduces it): generator, multiscale discriminator, temporal discriminator.  Behaviour the reference has and this class
keeps on purpose (all pinned by tests/golden/sams_*.npz, generated from the reference itself):

  * `generate_n_frames` runs the generator once per frame; previous frames enter detached, in the rotated order
    (f+1 .. f+n-1) mod n, together with the encoder label maps enc[:, n-1-f : -1] (sams_model.py:241-272);
  * with --flow_warp the raw RGB output is blended with the previous generated frame warped by the flow, using the
    raw 4th output channel as the weight — no tanh, no sigmoid (sams_model.py:229-237);
  * the generator's adversarial terms are evaluated on the discriminators' prediction for the REAL frames
    (sams_model.py:299-303,334-338); the temporal discriminator's loss averages over all of its stage outputs;
  * the discriminator steps regenerate the frames under no_grad with the generator still in training mode (running
    statistics and the spectral-norm vectors advance), and the temporal step reuses those frames.
"""
import argparse
import logging

import torch

from . import gan_options, ops
from . import tryon_channels as tc
from .base_model import BaseModel
from .networks.discriminator import MultiscaleDiscriminator, NLayerDiscriminator
from .networks.loss import GANLoss, VGGLoss
from .networks.sams.sams_generator import SamsGenerator
from .networks.sams.spade import stacked_weight_cache
from .optim import HipAdam
from .pl_compat import EvalResult, TrainResult
from .unet_mask_model import Resample2d

logger = logging.getLogger("logger")


class SamsModel(BaseModel):
    """Self Attentive Multi-Spade"""

    @classmethod
    def modify_commandline_options(cls, parser, is_train):
        parser = argparse.ArgumentParser(parents=[parser], add_help=False)
        parser = super(SamsModel, cls).modify_commandline_options(parser, is_train)
        parser.set_defaults(person_inputs=("agnostic", "densepose", "flow"))
        parser.add_argument("--encoder_input", default="flow",
                            help="which of the --person_inputs to use as the encoder segmap input (only 1 allowed).")
        parser.set_defaults(n_frames_total=5)
        parser.set_defaults(batch_size=4)
        parser.add_argument("--wt_l1", type=float, default=1.0, help="Weight applied to l1 loss in the generator")
        parser.add_argument("--wt_vgg", type=float, default=1.0, help="Weight applied to vgg loss in the generator")
        parser.add_argument("--wt_multiscale", type=float, default=1.0,
                            help="Weight applied to adversarial multiscale loss in the generator")
        parser.add_argument("--wt_temporal", type=float, default=1.0,
                            help="Weight applied to adversarial temporal loss in the generator")
        parser.add_argument("--norm_D", type=str, default="spectralinstance",
                            help="instance normalization or batch normalization")
        parser.add_argument("--vgg_weights", default=None, help="torchvision-format vgg19 state_dict file (.pth)")
        parser.add_argument("--allow_random_vgg", action="store_true",
                            help="train with frozen RANDOM VGG weights when no pretrained file is available")
        parser = SamsGenerator.modify_commandline_options(parser, is_train)
        if is_train:
            parser = MultiscaleDiscriminator.modify_commandline_options(parser, is_train)
        parser = gan_options.modify_commandline_options(parser, is_train)
        return parser

    @staticmethod
    def apply_default_encoder_input(opt):
        if hasattr(opt, "encoder_input") and opt.encoder_input is None:
            opt.encoder_input = opt.person_inputs[0]
        return opt

    def __init__(self, hparams):
        if isinstance(hparams, dict):
            hparams = argparse.Namespace(**hparams)
        super().__init__(hparams)
        self.n_frames_total = hparams.n_frames_total
        self.n_frames_now = hparams.n_frames_now if getattr(hparams, "n_frames_now", None) else self.n_frames_total
        self.inputs = list(hparams.person_inputs) + list(hparams.cloth_inputs)
        self.generator = SamsGenerator(hparams)
        self.resample = Resample2d()
        self._frames = None
        if self.is_train:
            init = hparams.init_type, hparams.init_variance
            self.generator.init_weights(*init)
            self.multiscale_discriminator = MultiscaleDiscriminator(hparams)
            self.multiscale_discriminator.init_weights(*init)
            enc_ch = tc.parse_num_channels(hparams.encoder_input)
            self.temporal_discriminator = NLayerDiscriminator(hparams, in_channels=self.n_frames_total * (enc_ch + tc.RGB_CHANNELS))
            self.temporal_discriminator.init_weights(*init)
            self.criterion_GAN = GANLoss(hparams.gan_mode)
            self.criterion_VGG = VGGLoss(weights_file=getattr(hparams, "vgg_weights", None))
            self.wt_l1, self.wt_vgg = hparams.wt_l1, hparams.wt_vgg
            self.wt_multiscale, self.wt_temporal = hparams.wt_multiscale, hparams.wt_temporal

    # ---- optimisation ----------------------------------------------------------------------------
    def optimizer_networks(self):
        """The networks whose parameters optimizer 0, 1, 2 own (sams_model.py:130-138)."""
        return [self.generator, self.multiscale_discriminator, self.temporal_discriminator]

    def configure_optimizers(self):
        hp = self.hparams
        opts = [HipAdam(list(net.parameters()), lr) for net, lr in zip(self.optimizer_networks(), (hp.lr, hp.lr_D, hp.lr_D))]
        return opts, [self._make_step_scheduler(o) for o in opts]

    def batch_keys(self):
        """Tensor entries of the batch this model reads: training (image, the label-map input, the person inputs, the
        optical flow of generate_n_frames under --flow_warp) and what the reference's visualize() shows (cloth + inputs)."""
        keys = {"image", "cloth", self.hparams.encoder_input, *self.inputs}
        if getattr(self.hparams, "flow_warp", False):
            keys.add("flow")
        return sorted(keys)

    def require_pretrained_vgg(self):
        """Called by Trainer.fit before training: the reference trains against ImageNet VGG19 features
        (models/networks/vgg.py:9); random ones only when explicitly allowed (synthetic benchmarks, tests)."""
        if not self.criterion_VGG.vgg.pretrained_loaded and not getattr(self.hparams, "allow_random_vgg", False):
            raise RuntimeError("SamsModel: no pretrained VGG19 weights for the perceptual loss (torchvision is not installed / "
                               "offline). Pass --vgg_weights <vgg19 state_dict or checkpoint>, or --allow_random_vgg.")

    # ---- frames ------------------------------------------------------------------------------------
    @property
    def all_gen_frames(self):
        """(b, n, 3, h, w) planar copy of the generated frames (visualisation / tests)."""
        if self._frames is None:
            return None
        ref = next(f for f in self._frames if f is not None)
        planes = [ops.to_nchw(f.detach()) if f is not None else None for f in self._frames]
        blank = torch.zeros_like(next(p for p in planes if p is not None)) if ref is not None else None
        return torch.stack([p if p is not None else blank for p in planes], 1)  # progressive training: unused = zeros

    def forward(self, *args, **kwargs):
        return self.generator(*args, **kwargs)

    def training_step(self, batch, batch_idx, optimizer_idx=0):
        if optimizer_idx == 0:
            return self.generator_step(batch)
        if optimizer_idx == 1:
            return self.multiscale_discriminator_step(batch)
        result = self.temporal_discriminator_step(batch)
        if self.global_step % self.hparams.display_count == 0:
            self.visualize(batch)
        return result

    def validation_step(self, batch, idx):
        self.batch = batch
        result = self.generator_step(batch, val=True)
        result.global_step = self.global_step
        return result

    def test_step(self, *args, **kwargs):
        """The reference leaves SAMS inference unimplemented (`pass`, sams_model.py:170-171)."""

    def _previous_inputs(self, batch, f, frames, zero):
        """Channel-stacked previous frames (detached, rotated order) and previous encoder label maps."""
        n = self.n_frames_total
        if n == 1:
            raise IndexError("SamsModel needs n_frames_total > 1 (models/sams_model.py:220 indexes a frames axis)")
        order = [(i + 1) % n for i in range(f, f + n - 1)]
        prev_frames = ops.cat_channels([frames[i].detach() if frames[i] is not None else zero for i in order])
        enc = batch[self.hparams.encoder_input]
        b, _, c, h, w = enc.shape
        start = n - 1 - f
        prev_maps = torch.cat((enc.new_zeros(b, start, c, h, w), enc[:, start:-1]), 1).reshape(b, -1, h, w)
        return prev_frames, prev_maps

    def generate_n_frames(self, batch):
        """-> (last generated frame, label maps of the last generated frame index, list of n frames (None = unused))."""
        hp = self.hparams
        image = batch["image"]
        b, n, _, h, w = image.shape
        zero = ops.fill_(ops.nhwc_empty(b, h, w, tc.RGB_CHANNELS, image.device), 0.0)
        frames = [None] * n
        fake = maps_now = None
        with stacked_weight_cache():  # the weights do not change between the passes of one generation
            for f in range(n - self.n_frames_now, n):
                maps_now = {k: ops.to_rows(batch[k][:, f]) for k in self.inputs}
                prev_frames, prev_maps = self._previous_inputs(batch, f, frames, zero)
                out = self.generator(prev_frames, prev_maps, maps_now)
                if hp.flow_warp:
                    last = frames[f - 1] if f > 0 and frames[f - 1] is not None else zero
                    warped = self.resample(last, batch["flow"][:, f].contiguous())
                    fake = ops.blend(warped, out[:, :3], out[:, 3:])  # (1 - m) * warped + m * rgb
                else:
                    fake = out[:, :3] if out.shape[1] > 3 else out
                frames[f] = fake
        return fake, maps_now, frames

    def mask_unused_frames(self, tensor):
        n_mask = self.n_frames_total - self.n_frames_now
        return torch.cat((torch.zeros_like(tensor[:, :n_mask]), tensor[:, n_mask:]), 1)

    def discriminate(self, discriminator, input_semantics, fake_image, real_image):
        """Fake and real go through the discriminator as ONE batch (sams_model.py:385-403)."""
        fake_concat = ops.cat_channels([input_semantics, fake_image])
        real_concat = ops.cat_channels([input_semantics, real_image])
        out = discriminator(ops.cat_batch([fake_concat, real_concat]))
        return split_predictions(out)

    def _adversarial(self, pred_fake, pred_real, for_discriminator):
        loss_real = self.criterion_GAN(pred_real, True, for_discriminator=for_discriminator)
        if not for_discriminator:
            return loss_real
        loss_fake = self.criterion_GAN(pred_fake, False, for_discriminator=for_discriminator)
        return (loss_fake + loss_real) / 2, loss_real, loss_fake

    def multiscale_adversarial_loss(self, batch, for_discriminator):
        if not for_discriminator:
            fake, maps_now, frames = self.generate_n_frames(batch)
        else:
            with torch.no_grad():
                fake, maps_now, frames = self.generate_n_frames(batch)
        self._frames = frames
        semantics = ops.cat_channels(list(maps_now.values()))
        truth = ops.to_rows(batch["image"][:, -1])
        pred_fake, pred_real = self.discriminate(self.multiscale_discriminator, semantics, fake, truth)
        return self._adversarial(pred_fake, pred_real, for_discriminator)

    def temporal_adversarial_loss(self, batch, for_discriminator):
        image = batch["image"]
        b, n, _, h, w = image.shape
        reals = ops.to_rows(self.mask_unused_frames(image).reshape(b, -1, h, w))
        zero = None
        parts = []
        for f in self._frames:
            if f is None:
                zero = zero if zero is not None else ops.fill_(ops.nhwc_empty(b, h, w, tc.RGB_CHANNELS, image.device), 0.0)
                f = zero
            parts.append(f if not for_discriminator else f.detach())
        fakes = ops.cat_channels(parts)
        semantics = ops.to_rows(self.mask_unused_frames(batch[self.hparams.encoder_input]).reshape(b, -1, h, w))
        pred_fake, pred_real = self.discriminate(self.temporal_discriminator, semantics, fakes, reals)
        return self._adversarial(pred_fake, pred_real, for_discriminator)

    # ---- the three steps ---------------------------------------------------------------------------
    def generator_step(self, batch, val=False):
        adv_multi = self.multiscale_adversarial_loss(batch, for_discriminator=False) * self.wt_multiscale
        adv_temporal = self.temporal_adversarial_loss(batch, for_discriminator=False) * self.wt_temporal
        truth = ops.to_rows(batch["image"][:, -1])
        fake = self._frames[-1]
        loss_l1 = ops.l1_loss(fake, truth) * self.wt_l1
        loss_vgg = self.criterion_VGG(fake, truth) * self.wt_vgg
        loss = loss_l1 + loss_vgg + adv_multi + adv_temporal
        v = "val_" if val else ""
        result = EvalResult(checkpoint_on=loss_l1 + loss_vgg) if val else TrainResult(loss)
        result.log(f"{v}loss", loss)
        result.log(f"{v}loss/G/adv_multiscale", adv_multi, prog_bar=True)
        result.log(f"{v}loss/G/adv_temporal", adv_temporal, prog_bar=True)
        result.log(f"{v}loss/G/l1+vgg", loss_l1 + loss_vgg)
        result.log(f"{v}loss/G/l1", loss_l1)
        result.log(f"{v}loss/G/vgg", loss_vgg)
        return result

    def multiscale_discriminator_step(self, batch):
        loss, real, fake = self.multiscale_adversarial_loss(batch, for_discriminator=True)
        result = TrainResult(loss)
        result.log("loss/D/multi", loss, prog_bar=True)
        result.log("loss/D/multi_fake", fake)
        result.log("loss/D/multi_real", real)
        return result

    def temporal_discriminator_step(self, batch):
        loss, real, fake = self.temporal_adversarial_loss(batch, for_discriminator=True)
        result = TrainResult(loss)
        result.log("loss/D/temporal", loss, prog_bar=True)
        result.log("loss/D/temporal_fake", fake)
        result.log("loss/D/temporal_real", real)
        return result


def split_predictions(pred):
    """First half of the batch = fake, second half = real, at every nesting level the discriminators return."""
    def halves(t):
        k = t.size(0) // 2
        return t[:k], t[k:]

    if isinstance(pred, list):
        fake, real = [], []
        for p in pred:
            if isinstance(p, torch.Tensor):
                a, b = halves(p)
            else:
                pairs = [halves(t) for t in p]
                a, b = [x for x, _ in pairs], [y for _, y in pairs]
            fake.append(a)
            real.append(b)
        return fake, real
    return halves(pred)

"""Generator of the Self-Attentive Multi-SPADE GAN (reference: models/networks/sams/sams_generator.py).

Encoder: 3x3 conv on the previous n-1 generated frames (stacked on channels), then [SPADE residual block, x0.5 nearest]
per power of `ngf_base`, conditioned on the previous frames' encoder label map.  Middle: `num_middle` channel-preserving
(Attentive)MultiSpade blocks conditioned on ALL label maps of the current frame.  Decoder: [x2 nearest, block] back to
`ngf_base ** ngf_pow_outer` features and a 3x3 conv to RGB (+ 1 blend-mask channel with --flow_warp).
The module lists (`encode_layers`, `middle_layers`, `decode_layers`) hold the same entries at the same indices as the
reference, so checkpoints are interchangeable.
"""
import logging
import sys

from torch import nn

from ... import ops_sams
from ... import tryon_channels as tc
from ..base_network import BaseNetwork
from ..layers import HipConv2d
from .attentive_multispade import AttentiveMultiSpade
from .multispade import MultiSpade
from .spade import SPADE, AnySpadeResBlock, label_pyramid

logger = logging.getLogger("logger")


class NearestResize(nn.Module):
    """nn.Upsample(scale_factor=s) with its default mode "nearest"; s = 0.5 shrinks (sams_generator.py:294-308)."""

    def __init__(self, scale_factor):
        super().__init__()
        self.scale_factor = scale_factor

    def forward(self, x):
        return ops_sams.resize_nearest(x, scale_factor=self.scale_factor)

    def extra_repr(self):
        return f"scale_factor={self.scale_factor}, mode=nearest"


def choose_spade_class_by_index(attn_indices, i, total_layers):
    """Attention where the layer's positive OR negative index is listed (as strings, straight from argparse)."""
    return AttentiveMultiSpade if (str(i) in attn_indices or str(i - total_layers) in attn_indices) else MultiSpade


class SamsGenerator(BaseNetwork):
    @classmethod
    def modify_commandline_options(cls, parser, is_train):
        parser = BaseNetwork.modify_commandline_options(parser, is_train)
        parser.add_argument("--norm_G", default="spectralspadesyncbatch3x3")
        parser.add_argument("--ngf_base", type=int, default=2, help="Control the size of the network. ngf_base ** pow")
        parser.add_argument("--ngf_power_start", "--ngf_pow_outer", dest="ngf_pow_outer", type=int, default=6,
                            help="number of features at the outer ends = ngf_base ** ngf_pow_outer")
        parser.add_argument("--ngf_power_end", "--ngf_pow_inner", dest="ngf_pow_inner", type=int, default=10,
                            help="INCLUSIVE! number of features in the middle of the network = ngf_base ** ngf_pow_inner")
        parser.add_argument("--ngf_pow_step", type=int, default=1,
                            help="increment the power this much between layers until >= ngf_pow_inner")
        parser.add_argument("--num_middle", type=int, default=3,
                            help="Number of channel-preserving layers between the encoder and decoder")
        parser.add_argument("--attention_middle_indices", nargs="?", default=[], help="middle layer indices for attention")
        parser.add_argument("--attention_decoder_indices", nargs="?", default=[], help="decoder layer indices for attention")
        if "--ngf" in sys.argv:
            logger.warning("SamsGenerator does NOT use --ngf. Use --ngf_base, --ngf_pow_outer, --ngf_pow_inner, "
                           "--ngf_pow_step, and --num_middle to control the architecture.")
        return parser

    def __init__(self, hparams):
        super().__init__()
        assert hparams.ngf_base > 1, f"{hparams.ngf_base}"
        assert hparams.ngf_pow_inner >= 1, f"ngf_pow_inner={hparams.ngf_pow_inner}"
        self.hparams = hparams
        self.inputs = list(hparams.person_inputs) + list(hparams.cloth_inputs)
        num_prev = max(hparams.n_frames_total - 1, 1)
        self.in_channels = tc.RGB_CHANNELS * num_prev
        out_channels = tc.RGB_CHANNELS + (tc.MASK_CHANNELS if hparams.flow_warp else 0)
        base, step = hparams.ngf_base, hparams.ngf_pow_step
        outer, inner = int(base ** hparams.ngf_pow_outer), int(base ** hparams.ngf_pow_inner)

        def block(fin, fout, labels, spade_class):
            return AnySpadeResBlock(fin, fout, hparams.norm_G, labels, spade_class, hparams.activation)

        # ---- encoder: plain SPADE on the previous frames' encoder label map
        enc_labels = tc.parse_num_channels(hparams.encoder_input) * num_prev
        encode = [HipConv2d(self.in_channels, outer, 3, padding=1)]
        feat = outer
        for p in range(hparams.ngf_pow_outer, hparams.ngf_pow_inner, step):
            fin, feat = int(base ** p), int(base ** (p + step))
            encode += [block(fin, feat, enc_labels, SPADE), NearestResize(0.5)]
        if feat != inner:
            logger.warning(f"Final out_feat={feat} in encoder layers didn't match NGF_INNER={inner}, adding an extra layer.")
            encode += [block(feat, inner, enc_labels, SPADE), NearestResize(0.5)]
        self.encode_layers = nn.ModuleList(encode)

        # ---- middle / decoder: one SPADE per label map of the current frame
        labels = {name: tc.parse_num_channels(name) for name in sorted(self.inputs)}
        self.middle_layers = nn.ModuleList(
            block(inner, inner, labels, choose_spade_class_by_index(hparams.attention_middle_indices, i, hparams.num_middle))
            for i in range(hparams.num_middle))
        decode = []
        pows = range(hparams.ngf_pow_inner, hparams.ngf_pow_outer, -step)
        for i, p in enumerate(pows):
            fin, feat = int(base ** p), int(base ** (p - step))
            decode += [NearestResize(2), block(fin, feat, labels, choose_spade_class_by_index(hparams.attention_decoder_indices, i, len(pows)))]
        if feat != outer:
            logger.warning(f"Final out_feat={feat} in decoder layers didn't match NGF_OUTER={outer}, adding an extra layer.")
            extra = AttentiveMultiSpade if hparams.attention_decoder_indices else MultiSpade
            decode += [NearestResize(2), block(feat, outer, labels, extra)]
        decode.append(HipConv2d(outer, out_channels, 3, padding=1))
        self.decode_layers = nn.ModuleList(decode)

    def forward(self, prev_n_frames_G, prev_n_labelmaps, current_labelmap_dict):
        """prev_n_frames_G, prev_n_labelmaps: (b, n-1, c, h, w) as in the reference, or already stacked on the channel
        axis (b, (n-1) * c, h, w) — what SamsModel hands over; current_labelmap_dict: name -> (b, c, h, w)."""
        if self.hparams.n_frames_total > 1:
            x, prev_maps = prev_n_frames_G, prev_n_labelmaps
            if x.dim() == 5:
                x = x.reshape(x.shape[0], -1, *x.shape[3:])
            if prev_maps.dim() == 5:
                prev_maps = prev_maps.reshape(prev_maps.shape[0], -1, *prev_maps.shape[3:])
        else:
            raise IndexError("SamsGenerator needs n_frames_total > 1: the reference's SamsModel indexes a frames axis that "
                             "single-frame batches do not have (models/sams_model.py:220)")
        with label_pyramid():  # each label map is resized to each resolution once per pass
            for layer in self.encode_layers:
                x = layer(x, prev_maps) if isinstance(layer, AnySpadeResBlock) else layer(x)
            for layer in self.middle_layers:
                x = layer(x, current_labelmap_dict)
            for layer in self.decode_layers:
                x = layer(x, current_labelmap_dict) if isinstance(layer, AnySpadeResBlock) else layer(x)
        return x

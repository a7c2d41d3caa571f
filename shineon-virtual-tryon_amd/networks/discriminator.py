"""PatchGAN discriminators of the SAMS-GAN (reference: models/networks/discriminator.py).

NLayerDiscriminator: 4x4 convolutions with padding 2 — stride 2 except the last two — `model0` (conv + LeakyReLU(0.2)),
`model1..n-1` (norm_D(conv) + LeakyReLU(0.2)), `model<n>` (conv to 1 channel).  Its forward returns every stage's
output unless --no_ganFeat_loss.  MultiscaleDiscriminator runs `num_D` of them, average-pooling the input between scales.
LeakyReLU runs in the convolution epilogue whenever the conv is directly followed by it.
"""
import math

from torch import nn

from .. import ops, ops_sams
from .. import tryon_channels as tc
from .base_network import BaseNetwork
from .layers import HipConv2d, HipLeakyReLU
from .normalization import get_nonspade_norm_layer


class NLayerDiscriminator(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--n_layers_D", type=int, default=4, help="# layers in each discriminator")
        parser.add_argument("--ndf", type=int, default=64, help="num discriminator features")
        return parser

    def __init__(self, opt, in_channels=None):
        super().__init__()
        self.opt = opt
        kw = 4
        padw = int(math.ceil((kw - 1.0) / 2))
        nf = opt.ndf
        input_nc = in_channels if in_channels else self.compute_D_input_nc(opt)
        norm_layer = get_nonspade_norm_layer(opt, opt.norm_D)
        stages = [[HipConv2d(input_nc, nf, kw, stride=2, padding=padw, fuse_leaky=0.2), _Fused()]]
        for n in range(1, opt.n_layers_D):
            nf_prev, nf = nf, min(nf * 2, 512)
            stride = 1 if n == opt.n_layers_D - 1 else 2
            wrapped = norm_layer(HipConv2d(nf_prev, nf, kw, stride=stride, padding=padw))
            if isinstance(wrapped, HipConv2d):  # spectral norm only: the activation can ride in the epilogue
                wrapped.fused_act, wrapped.fused_act_param = ops.ACT_LEAKY, 0.2
                stages.append([wrapped, _Fused()])
            else:
                stages.append([wrapped, HipLeakyReLU(0.2)])
        stages.append([HipConv2d(nf, 1, kw, stride=1, padding=padw)])
        for n, stage in enumerate(stages):
            self.add_module("model" + str(n), nn.Sequential(*stage))

    def compute_D_input_nc(self, opt):
        return tc.parse_num_channels(opt.person_inputs) + tc.parse_num_channels(opt.cloth_inputs) + tc.RGB_CHANNELS

    def forward(self, input):
        results = [input]
        for stage in self.children():
            results.append(stage(results[-1]))
        return results[1:] if not self.opt.no_ganFeat_loss else results[-1]


class _Fused(nn.Module):
    """Keeps the nn.Sequential slot of a LeakyReLU(0.2) that the preceding convolution applies in its epilogue."""

    def forward(self, x):
        return x


class MultiscaleDiscriminator(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--netD_subarch", type=str, default="n_layer", help="architecture of each discriminator")
        parser.add_argument("--num_D", type=int, default=2, help="number of discriminators to be used in multiscale")
        opt, _ = parser.parse_known_args()
        if opt.netD_subarch != "n_layer":
            raise ValueError("unrecognized discriminator subarchitecture %s" % opt.netD_subarch)
        NLayerDiscriminator.modify_commandline_options(parser, is_train)
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        for i in range(opt.num_D):
            if opt.netD_subarch != "n_layer":
                raise ValueError("unrecognized discriminator subarchitecture %s" % opt.netD_subarch)
            self.add_module("discriminator_%d" % i, NLayerDiscriminator(opt))

    def downsample(self, input):
        return ops_sams.avg_pool3s2(input)

    def forward(self, input):
        """-> list (one entry per scale) of lists (that discriminator's stage outputs)."""
        result = []
        keep_features = not self.opt.no_ganFeat_loss
        children = list(self.children())
        for k, D in enumerate(children):
            out = D(input)
            result.append(out if keep_features else [out])
            if k + 1 < len(children):  # the reference also pools after the last scale; that result is never used
                input = self.downsample(input)
        return result

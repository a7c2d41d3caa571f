"""SPADE and the SPADE residual block of the SAMS generator (reference: models/networks/sams/spade.py).

Module tree (state_dict keys) as in the reference: `param_free_norm`, `mlp_shared.0`, `mlp_gamma`, `mlp_beta`;
`conv_0 / conv_1 / conv_s`, `spade_0 / spade_1 / norm_s`.  Execution differs: mlp_gamma and mlp_beta read the same
activation, so they run as ONE convolution with stacked weights, and the modulation `n * (1 + gamma) + beta` is a
single pass that also applies the residual block's activation when one follows.
"""
import re

import torch
from torch import nn

from ... import ops, ops_sams
from ..layers import HipBatchNorm2d, HipConv2d, HipInstanceNorm2d
from ..spectral import spectral_norm
from ..sync_batchnorm import SynchronizedBatchNorm2d

# activation names -> op codes.  The SPADE MLP reads "relu" as a true ReLU (spade.py:93-103), the residual block as
# LeakyReLU(0.2) (spade.py:182-192).
_MLP_ACT = {"relu": ("relu", 0.0), "gelu": ("gelu", 0.0), "swish": ("swish", 0.0), "sine": ("sine", 0.0)}
_BLOCK_ACT = {"relu": ("leaky", 0.2), "gelu": ("gelu", 0.0), "swish": ("swish", 0.0), "sine": ("sine", 0.0)}


class stacked_weight_cache:
    """Context manager: inside it every SPADE stacks its mlp_gamma / mlp_beta parameters ONCE and reuses the result.
    SamsModel.generate_n_frames runs the generator once per frame with the same weights, so the 84 SPADEs of the default
    generator stack 5x less often; under autograd the passes share one node, whose gradient is split back into the two
    parameters once.  Nothing is cached outside the context (an optimizer step in between would make it stale)."""

    active = None

    def __enter__(self):
        self.prev = stacked_weight_cache.active
        stacked_weight_cache.active = {}
        return self

    def __exit__(self, *exc):
        stacked_weight_cache.active = self.prev
        return False


class label_pyramid:
    """Context manager around ONE generator pass: a label map resized to a resolution (and zero-padded to the 4-channel
    granularity of the convolution loaders) is computed once and shared by every SPADE that conditions on it at that
    resolution - 4 maps x 5 resolutions instead of one resize + one pad per SPADE (84 in the default generator)."""

    active = None

    def __enter__(self):
        self.prev = label_pyramid.active
        label_pyramid.active = {}
        return self

    def __exit__(self, *exc):
        label_pyramid.active = self.prev
        return False

    @staticmethod
    def resized(segmap, size):
        cache = label_pyramid.active
        key = (segmap.data_ptr(), tuple(segmap.shape), tuple(int(s) for s in size))
        if cache is not None and key in cache:
            return cache[key][1]
        seg = ops_sams.resize_nearest(segmap, size=size)
        c = seg.shape[1]
        if c % 4:
            cp = (c + 3) // 4 * 4
            seg = ops.to_rows(seg, cpad=cp)[:, :c]
            seg._so_zero_padded = cp  # ops.conv2d then reads the padded buffer in place instead of padding its own copy
        if cache is not None:
            cache[key] = (segmap, seg)  # the source tensor is kept alive so its address cannot be recycled inside the pass
        return seg


def _lookup(table, name):
    if name not in table:
        raise RuntimeError(f"The selected activation should be relu/gelu/swish/sine, not {name}")
    return table[name]


class SPADE(nn.Module):
    @staticmethod
    def parse_config_text(config_text):
        """"spade<norm><k>x<k>" -> (parameter-free norm class, kernel size) (spade.py:36-59)."""
        assert config_text.startswith("spade")
        parsed = re.search(r"spade(\D+)(\d)x\d", config_text)
        kind = str(parsed.group(1))
        norms = {"instance": HipInstanceNorm2d, "syncbatch": SynchronizedBatchNorm2d, "batch": HipBatchNorm2d}
        if kind not in norms:
            raise ValueError("%s is not a recognized param-free norm type in SPADE" % kind)
        return norms[kind], int(parsed.group(2))

    def __init__(self, config_text, norm_nc, label_nc, activation):
        super().__init__()
        norm_cls, ks = SPADE.parse_config_text(config_text)
        self.param_free_norm = norm_cls(norm_nc) if norm_cls is HipInstanceNorm2d else norm_cls(norm_nc, affine=False)
        self.mlp_act = _lookup(_MLP_ACT, activation)
        self.nhidden = nhidden = 128  # hard-coded in the reference (spade.py:68)
        pw = ks // 2
        fuse = self.mlp_act[0] == "relu"
        # index 0 keeps the reference's nn.Sequential(conv, activation) key `mlp_shared.0`
        self.mlp_shared = nn.ModuleList([HipConv2d(label_nc, nhidden, ks, padding=pw, fuse_relu=fuse)])
        self.mlp_gamma = HipConv2d(nhidden, norm_nc, ks, padding=pw)
        self.mlp_beta = HipConv2d(nhidden, norm_nc, ks, padding=pw)

    def forward(self, x, segmap, then_act=None):
        """then_act: (kind, param) of an activation applied to the result in the same pass."""
        seg = label_pyramid.resized(segmap, x.shape[2:])
        actv = self.mlp_shared[0](seg)
        if self.mlp_act[0] != "relu":
            actv = ops.activation(actv, *self.mlp_act)
        cache = stacked_weight_cache.active
        key = (id(self), torch.is_grad_enabled())
        if cache is not None and key in cache:
            w2, b2 = cache[key]
        else:
            w2, b2 = ops_sams.stack_conv_params(self.mlp_gamma.weight, self.mlp_gamma.bias, self.mlp_beta.weight, self.mlp_beta.bias)
            if cache is not None:
                cache[key] = (w2, b2)
        gamma_beta = ops.conv2d(actv, w2, b2, 1, self.mlp_gamma.padding, bias_grad_hint=True)
        kind, param = then_act if then_act is not None else ("none", 0.0)
        norm = self.param_free_norm
        if isinstance(norm, HipInstanceNorm2d):
            return ops_sams.spade_norm_modulate(x, gamma_beta, instance=True, eps=norm.eps, act=kind, param=param)
        if self.training:  # batch statistics: normalisation, modulation and activation in one pass over x
            if norm.count_batches and not norm._shared_counter:
                norm.num_batches_tracked += 1
            return ops_sams.spade_norm_modulate(x, gamma_beta, norm.running_mean, norm.running_var, instance=False,
                                                momentum=norm.momentum, eps=norm.eps, act=kind, param=param)
        return ops_sams.spade_modulate(norm(x), gamma_beta, kind, param)  # eval: running statistics


class AnySpadeResBlock(nn.Module):
    """norm -> activation -> conv, twice, plus a (learned, if the channel count changes) shortcut (spade.py:106-180)."""

    def __init__(self, fin, fout, norm_G, label_channels, spade_class, activation):
        super().__init__()
        self.learned_shortcut = fin != fout
        fmiddle = min(fin, fout)
        self.conv_0 = HipConv2d(fin, fmiddle, 3, padding=1)
        self.conv_1 = HipConv2d(fmiddle, fout, 3, padding=1)
        if self.learned_shortcut:
            self.conv_s = HipConv2d(fin, fout, 1, padding=0, bias=False)
        if "spectral" in norm_G:
            self.conv_0 = spectral_norm(self.conv_0)
            self.conv_1 = spectral_norm(self.conv_1)
            if self.learned_shortcut:
                self.conv_s = spectral_norm(self.conv_s)
        config = norm_G.replace("spectral", "")
        self.spade_0 = spade_class(config, fin, label_channels, activation)
        self.spade_1 = spade_class(config, fmiddle, label_channels, activation)
        if self.learned_shortcut:
            self.norm_s = spade_class(config, fin, label_channels, activation)
        self.block_act = _lookup(_BLOCK_ACT, activation)

    def forward(self, x, seg):
        x_s = self.conv_s(self.norm_s(x, seg)) if self.learned_shortcut else x
        dx = self.conv_0(self.spade_0(x, seg, then_act=self.block_act))
        dx = self.conv_1(self.spade_1(dx, seg, then_act=self.block_act))
        return ops.add(x_s, dx)

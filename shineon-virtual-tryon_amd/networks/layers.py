"""nn.Module shells around the HIP operators.

They exist so that the module TREE (attribute names, nn.Sequential indices) — and therefore every
`state_dict` key and tensor shape — is identical to the reference's, while the
arithmetic runs in libshineon_hip.so.  Class names keep the substrings `Conv`, `Linear`,
`BatchNorm2d` because the reference's `weights_init_normal` dispatches on them
(models/networks/__init__.py:52-60).
"""
import math

import torch
from torch import nn
from torch.nn import init

from .. import ops


def _ohwi_param(o, i, r, s):
    """(O, I, R, S) parameter whose memory is OHWI (what the MFMA loaders read in place)."""
    return nn.Parameter(torch.empty(o, r, s, i).permute(0, 3, 1, 2))


class HipConv2d(nn.Module):
    """nn.Conv2d(in, out, k, stride, padding, bias) replacement (zero padding, dilation 1, groups 1)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, fuse_relu=False,
                 fuse_leaky=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.fuse_relu = fuse_relu
        # fuse_leaky = slope: LeakyReLU(slope) in the convolution's epilogue (PatchGAN layers, discriminator.py:108-111)
        self.fused_act = ops.ACT_RELU if fuse_relu else (ops.ACT_LEAKY if fuse_leaky is not None else ops.ACT_NONE)
        self.fused_act_param = float(fuse_leaky) if fuse_leaky is not None else 0.0
        # Set by the owning block when this conv feeds an Instance/BatchNorm directly: the bias gradient is then
        # analytically zero (the norm removes per-channel constants) and its reduction is skipped.
        # "always": InstanceNorm follows; "train": BatchNorm follows (true only with batch statistics).
        self.zero_bias_grad = None
        self.weight = _ohwi_param(out_channels, in_channels, kernel_size, kernel_size)
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # PyTorch's default Conv2d init (torch/nn/modules/conv.py): the reference keeps it for biases
        # everywhere and for all FeatureRegression weights.
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size * self.kernel_size
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        zbg = self.zero_bias_grad == "always" or (self.zero_bias_grad == "train" and self.training)
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.fused_act, zero_bias_grad=zbg,
                          act_param=self.fused_act_param)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, bias={self.bias is not None}, fuse_relu={self.fuse_relu}")


class HipInstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(C) with its defaults: affine=False, track_running_stats=False, eps=1e-5."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps

    def forward(self, x):
        return ops.instance_norm(x, self.eps)


class HipBatchNorm2d(nn.Module):
    """nn.BatchNorm2d(C, affine=...): batch statistics in training (running stats updated with momentum 0.1 and the
    unbiased variance), running statistics in eval.  count_batches=False reproduces a bare F.batch_norm call, which
    updates the running statistics but not num_batches_tracked (the reference's SynchronizedBatchNorm2d outside
    DataParallelWithCallback, models/networks/sync_batchnorm/batchnorm.py:63-68)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, count_batches=True):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.affine, self.count_batches = affine, count_batches
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._shared_counter = False  # True: the owner model bumps all counters with one launch (share_bn_counters)

    def forward(self, x, relu_gate_input=False, conv_bias=None):
        """relu_gate_input (training only): x is a ReLU output whose producer leaves the ReLU's backward mask to this
        layer (see ops.batch_norm_train); conv_bias: that producer's bias parameter - its gradient (the column sums of the
        gated input gradient) then comes out of this layer's backward statistics pass."""
        if self.training:
            if self.count_batches and not self._shared_counter:
                self.num_batches_tracked += 1
            return ops.batch_norm_train(x, self.weight, self.bias, self.running_mean, self.running_var,
                                        self.momentum, self.eps, relu_gate_input=relu_gate_input, conv_bias=conv_bias)
        return ops.batch_norm_eval(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps)


class HipLinearCHW(nn.Module):
    """nn.Linear applied to `x.view(N, -1)` of an (N, C, H, W) map, optionally followed by tanh.
    The weight keeps the reference's (C, H, W) flatten order (warp.py:87,95-97)."""

    def __init__(self, in_features, out_features, apply_tanh=False):
        super().__init__()
        self.in_features, self.out_features, self.apply_tanh = in_features, out_features, apply_tanh
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return ops.linear_chw_tanh(x, self.weight, self.bias, self.apply_tanh)


class HipActivation(nn.Module):
    def __init__(self, kind, param=0.0):
        super().__init__()
        self.kind, self.param = kind, param

    def forward(self, x):
        return ops.activation(x, self.kind, self.param)

    def extra_repr(self):
        return f"{self.kind}" + (f", {self.param}" if self.param else "")


class HipLeakyReLU(HipActivation):
    def __init__(self, negative_slope=0.2):
        super().__init__("leaky", negative_slope)


class HipReLU(HipActivation):
    def __init__(self):
        super().__init__("relu")


class HipGELU(HipActivation):
    def __init__(self):
        super().__init__("gelu")


class Swish(HipActivation):
    def __init__(self):
        super().__init__("swish")


class Sine(HipActivation):
    def __init__(self):
        super().__init__("sine")


class FusedReLU(nn.Module):
    """Placeholder keeping the nn.Sequential index of a ReLU whose work is fused in the preceding
    convolution's epilogue (VGG19 slices)."""

    def forward(self, x):
        return x


class HipUpsample2x(nn.Module):
    """nn.Upsample(scale_factor=2, mode="bilinear") (align_corners=False)."""

    def forward(self, x):
        return ops.upsample2x_bilinear(x)


class HipMaxPool2x2(nn.Module):
    def forward(self, x):
        return ops.maxpool2x2(x)


def share_bn_counters(model):
    """Re-home every HipBatchNorm2d.num_batches_tracked of `model` as a view of ONE int64 tensor and return it, so a
    training forward bumps all of them with a single `flat.add_(1)` instead of one tiny kernel per layer (14 in the
    GMM).  state_dict keys / values are unchanged (load_state_dict copies into the views in place)."""
    bns = [m for m in model.modules() if isinstance(m, HipBatchNorm2d)]
    if not bns:
        return None
    flat = torch.stack([m.num_batches_tracked.reshape(()) for m in bns]).contiguous()
    for i, m in enumerate(bns):
        m._buffers["num_batches_tracked"] = flat[i]
        m._shared_counter = True
    return flat

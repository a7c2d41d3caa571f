"""Geometric-matching networks (reference: models/networks/cpvton/warp.py:9-318).

FeatureExtraction (Conv -> ReLU -> BatchNorm order!), FeatureL2Norm, FeatureCorrelation,
FeatureRegression (Conv -> BatchNorm -> ReLU, Linear, tanh) and TpsGridGen, same constructors and
state_dict layout; the arithmetic is in libshineon_hip.so.
"""
import numpy as np
import torch
from torch import nn

from ... import ops
from .. import init_weights
from ..layers import HipBatchNorm2d, HipConv2d, HipLinearCHW, HipReLU


def _resolve_norm(norm_layer):
    if norm_layer in (HipBatchNorm2d, nn.BatchNorm2d, "batch"):
        return HipBatchNorm2d
    raise ValueError(f"unsupported norm_layer {norm_layer} (the reference only uses BatchNorm2d here)")


class FeatureExtraction(nn.Module):
    def __init__(self, input_nc, ngf=64, n_layers=3, norm_layer=HipBatchNorm2d, use_dropout=False):
        super().__init__()
        norm_layer = _resolve_norm(norm_layer)
        model = [HipConv2d(input_nc, ngf, kernel_size=4, stride=2, padding=1), HipReLU(), norm_layer(ngf)]
        for i in range(n_layers):
            in_ngf = 2 ** i * ngf if 2 ** i * ngf < 512 else 512
            out_ngf = 2 ** (i + 1) * ngf if 2 ** i * ngf < 512 else 512
            model += [HipConv2d(in_ngf, out_ngf, kernel_size=4, stride=2, padding=1), HipReLU()]
            model += [norm_layer(out_ngf)]
        model += [HipConv2d(512, 512, kernel_size=3, stride=1, padding=1), HipReLU()]
        model += [norm_layer(512)]
        model += [HipConv2d(512, 512, kernel_size=3, stride=1, padding=1), HipReLU()]
        self.model = nn.Sequential(*model)
        init_weights(self.model, init_type="normal")

    def forward(self, x):
        """nn.Sequential semantics with the Conv -> ReLU (-> BatchNorm) groups fused: the ReLU runs in the convolution's
        epilogue, and in training its backward mask is applied by the BatchNorm backward that already reads the
        same tensor (no separate activation kernels in either direction)."""
        mods = list(self.model)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, HipConv2d) and not m.fuse_relu and i + 1 < len(mods) and isinstance(mods[i + 1], HipReLU):
                bn = mods[i + 2] if i + 2 < len(mods) and isinstance(mods[i + 2], HipBatchNorm2d) else None
                gate = bn is not None and bn.training and torch.is_grad_enabled()
                # the convolution's bias gradient = column sums of the gradient the BatchNorm hands back: with the bias in the
                # optimizer's slab the BatchNorm's own backward statistics pass accumulates it (no column-sum launches)
                ext = gate and m.bias is not None and m.bias.requires_grad and ops._direct_grad_ok(m.bias, False)
                x = ops.conv2d(x, m.weight, m.bias, m.stride, m.padding, ops.ACT_RELU, act_grad_external=gate,
                               bias_grad_external=ext)
                if bn is not None:
                    x = bn(x, relu_gate_input=gate, conv_bias=m.bias if ext else None)
                i += 3 if bn is not None else 2
            else:
                x = m(x)
                i += 1
        return x


class FeatureL2Norm(nn.Module):
    def forward(self, feature, transpose_hw=False):
        return ops.feature_l2norm(feature, transpose_hw)


class FeatureCorrelation(nn.Module):
    """corr[b, ja*h + ia, i, j] = sum_c A[b, c, ia, ja] * B[b, c, i, j]   (warp.py:57-67).

    `a_is_transposed=True` means feature_A is already the (N, C, W, H) map produced by
    FeatureL2Norm(transpose_hw=True), which saves the separate transpose pass."""

    def forward(self, feature_A, feature_B, a_is_transposed=False):
        if not a_is_transposed:
            feature_A = ops.to_rows(feature_A.transpose(2, 3))
        return ops.feature_correlation(feature_A, feature_B)


class FeatureRegression(nn.Module):
    def __init__(self, input_nc=512, output_dim=6):
        super().__init__()
        self.conv = nn.Sequential(
            HipConv2d(input_nc, 512, kernel_size=4, stride=2, padding=1), HipBatchNorm2d(512), HipReLU(),
            HipConv2d(512, 256, kernel_size=4, stride=2, padding=1), HipBatchNorm2d(256), HipReLU(),
            HipConv2d(256, 128, kernel_size=3, stride=1, padding=1), HipBatchNorm2d(128), HipReLU(),
            HipConv2d(128, 64, kernel_size=3, stride=1, padding=1), HipBatchNorm2d(64), HipReLU(),
        )
        for conv_idx in (0, 3, 6, 9):  # Conv -> BatchNorm directly: zero bias gradient in training mode
            self.conv[conv_idx].zero_bias_grad = "train"
        self.linear = HipLinearCHW(64 * 4 * 3, output_dim, apply_tanh=True)  # tanh fused (warp.py:88,98)
        self.tanh = nn.Identity()

    def forward(self, x):
        return self.linear(self.conv(x))


def _axis(count):
    """`count` points from -1 to 1: float64 linspace rounded to fp32."""
    return torch.from_numpy(np.linspace(-1.0, 1.0, count)).float()


class TpsGridGen(nn.Module):
    """Thin-plate-spline sampling grid (warp.py:116-318).  Constants are plain attributes (not buffers),
    like the reference, so they are absent from checkpoints."""

    def __init__(self, out_h=256, out_w=192, use_regular_grid=True, grid_size=3, reg_factor=0):
        super().__init__()
        if not use_regular_grid:
            raise NotImplementedError("the reference only builds TpsGridGen with use_regular_grid=True")
        self.out_h, self.out_w, self.reg_factor = out_h, out_w, reg_factor
        self.N = grid_size * grid_size
        # Output sampling lattice and control-point lattice: float64 linspace rounded once to fp32, the values the
        # reference's np.linspace / np.meshgrid produce (warp.py:128-151); only the two axes are kept, the kernels
        # index them per pixel.  Control points are enumerated x-major: point k sits at (axis[k // g], axis[k % g]).
        self._gx = _axis(out_w)
        self._gy = _axis(out_h)
        ctrl = _axis(grid_size)
        self.P_X_base = ctrl.repeat_interleave(grid_size).reshape(-1, 1)
        self.P_Y_base = ctrl.repeat(grid_size).reshape(-1, 1)
        self.Li = self.compute_L_inverse(self.P_X_base, self.P_Y_base).unsqueeze(0)
        self._dev_consts = {}

    @staticmethod
    def compute_L_inverse(X, Y):
        """fp32 inverse of the TPS system matrix L = [[K, P], [P^T, 0]] with K_ij = U(|p_i - p_j|^2), U(r2) = r2 ln r2
        (U := 0 on the diagonal via r2 -> 1) and P = [1 | x | y]   (warp.py:169-189; same LAPACK inverse)."""
        pts = torch.cat([X.reshape(-1, 1), Y.reshape(-1, 1)], dim=1)
        n = pts.shape[0]
        r2 = (pts.unsqueeze(1) - pts.unsqueeze(0)).square().sum(dim=2)
        r2 = torch.where(r2 == 0, torch.ones_like(r2), r2)
        system = torch.zeros(n + 3, n + 3, dtype=pts.dtype)
        system[:n, :n] = r2 * r2.log()
        system[:n, n] = 1.0
        system[:n, n + 1:] = pts
        system[n:, :n] = system[:n, n:].t()
        return torch.linalg.inv(system)

    def _consts(self, device):
        key = str(device)
        if key not in self._dev_consts:
            self._dev_consts[key] = tuple(
                t.contiguous().to(device)
                for t in (self.Li[0], self.P_X_base.view(-1), self.P_Y_base.view(-1), self._gx, self._gy)
            )
        return self._dev_consts[key]

    def forward(self, theta):
        if theta.dim() != 2:
            theta = theta.reshape(theta.size(0), -1)
        return ops.tps_grid(theta, self._consts(theta.device), self.out_h, self.out_w, self.N)

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
                x = ops.conv2d(x, m.weight, m.bias, m.stride, m.padding, ops.ACT_RELU, act_grad_external=gate)
                if bn is not None:
                    x = bn(x, relu_gate_input=gate)
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


class TpsGridGen(nn.Module):
    """Thin-plate-spline sampling grid (warp.py:116-318).  Constants are plain attributes (not buffers),
    like the reference, so they are absent from checkpoints."""

    def __init__(self, out_h=256, out_w=192, use_regular_grid=True, grid_size=3, reg_factor=0):
        super().__init__()
        self.out_h, self.out_w = out_h, out_w
        self.reg_factor = reg_factor
        # regular output grid: float64 linspace -> fp32 (bit-exact with the reference's np.meshgrid rows)
        self._gx = torch.from_numpy(np.linspace(-1, 1, out_w)).float()
        self._gy = torch.from_numpy(np.linspace(-1, 1, out_h)).float()
        if not use_regular_grid:
            raise NotImplementedError("the reference only builds TpsGridGen with use_regular_grid=True")
        axis_coords = np.linspace(-1, 1, grid_size)
        self.N = grid_size * grid_size
        P_Y, P_X = np.meshgrid(axis_coords, axis_coords)  # P_X varies slowest (warp.py:142-145)
        P_X = torch.FloatTensor(np.reshape(P_X, (-1, 1)))
        P_Y = torch.FloatTensor(np.reshape(P_Y, (-1, 1)))
        self.P_X_base, self.P_Y_base = P_X.clone(), P_Y.clone()
        self.Li = self.compute_L_inverse(P_X, P_Y).unsqueeze(0)
        self._dev_consts = {}

    @staticmethod
    def compute_L_inverse(X, Y):
        """fp32 inverse of L = [[K, P], [P^T, 0]], K = r^2 log r^2 (warp.py:169-189)."""
        N = X.size(0)
        Xmat, Ymat = X.expand(N, N), Y.expand(N, N)
        d2 = torch.pow(Xmat - Xmat.transpose(0, 1), 2) + torch.pow(Ymat - Ymat.transpose(0, 1), 2)
        d2[d2 == 0] = 1
        K = torch.mul(d2, torch.log(d2))
        O = torch.ones(N, 1)
        Z = torch.zeros(3, 3)
        P = torch.cat((O, X, Y), 1)
        L = torch.cat((torch.cat((K, P), 1), torch.cat((P.transpose(0, 1), Z), 1)), 0)
        return torch.inverse(L)

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

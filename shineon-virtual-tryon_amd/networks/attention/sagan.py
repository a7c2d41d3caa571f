"""SAGAN self-attention (reference: models/networks/attention/sagan.py:5-54).

out = gamma * (V softmax(Q^T K)^T) + x with 1x1 convolutions for Q, K (C/8 channels) and V (C channels);
no 1/sqrt(d) scaling; gamma initialised to 0.  All matmuls run on fp32 MFMA (so_gemm_batched), the row
softmax is one wavefront per query.
"""
import torch
from torch import nn

from ... import ops
from ..layers import HipConv2d


class SelfAttention(nn.Module):
    def __init__(self, in_dim, activation=None):
        super().__init__()
        self.chanel_in = in_dim  # (sic) attribute name of the reference
        self.activation = activation  # accepted and ignored, like the reference (unet.py:150)
        self.query_conv = HipConv2d(in_dim, in_dim // 8, kernel_size=1)
        self.key_conv = HipConv2d(in_dim, in_dim // 8, kernel_size=1)
        self.value_conv = HipConv2d(in_dim, in_dim, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def adjacent_param_groups(self):
        """Layout hint for optim.HipAdam: q/k/v weights (and biases) adjacent in the flat slab = one [2d+C, C] matrix, so
        that one GEMM serves the three projections (ops._SelfAttentionQkvFn)."""
        convs = (self.query_conv, self.key_conv, self.value_conv)
        return [tuple(c.weight for c in convs), tuple(c.bias for c in convs)]

    def forward(self, x):
        return ops.self_attention(
            x, self.query_conv.weight, self.query_conv.bias, self.key_conv.weight, self.key_conv.bias,
            self.value_conv.weight, self.value_conv.bias, self.gamma,
        )

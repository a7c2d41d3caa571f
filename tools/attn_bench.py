"""Self-attention module forward+backward time per N (hipGraph replay), LDS-resident core vs GEMM + softmax composition.
    python tools/attn_bench.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shineon_virtual_tryon_amd  # noqa: E402,F401
from shineon_virtual_tryon_amd import ops  # noqa: E402
from shineon_virtual_tryon_amd.networks.attention.sagan import SelfAttention  # noqa: E402
from shineon_virtual_tryon_amd.optim import HipAdam  # noqa: E402

dev = torch.device("cuda", 0)
for hw in ((4, 3), (8, 6), (16, 12)):
    for fused in (False, True):
        ops.FUSED_ATTENTION = fused
        torch.manual_seed(0)
        sa = SelfAttention(512).to(dev)
        sa.gamma.data.fill_(0.5)
        opt = HipAdam(sa.parameters(), lr=1e-3, adjacent=sa.adjacent_param_groups())
        opt.zero_grad()
        x = torch.randn(4, 512, *hw, device=dev).requires_grad_(True)
        seed = torch.randn(4, 512, *hw, device=dev)

        def step():
            y = sa(x)
            (y * seed).sum().backward()

        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"N={hw[0] * hw[1]:4d} fused={int(fused)}: {e0.elapsed_time(e1) / 200 * 1e3:7.1f} us per fwd+bwd (incl. qkv GEMMs, seed mul/sum)")

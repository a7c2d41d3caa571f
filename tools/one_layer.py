"""Runs ONE layer-sized kernel repeatedly (for rocprofv3 --pmc / --kernel-trace passes of a single instantiation).

    python tools/one_layer.py wgrad 4 16 12 512 512 3 1 1        # mode N H W Cin Cout k stride pad
    python tools/one_layer.py wino  8 256 192 64 64 3 1 1         # fused Winograd F(2x2,3x3)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shineon_virtual_tryon_amd as pkg  # noqa: E402
from shineon_virtual_tryon_amd import ops  # noqa: E402


def main():
    mode = sys.argv[1]
    n, h, w, ci, co, k, s, p = (int(v) for v in sys.argv[2:10])
    reps = int(sys.argv[10]) if len(sys.argv) > 10 else 20
    L = pkg.lib()
    if os.environ.get("WINO_DMA") is not None:      # fused Winograd kernel: LDS-DMA staging (1, default) or registers (0)
        L.so_wino_fused_dma(int(os.environ["WINO_DMA"]))
    if os.environ.get("WINO_KBMAJOR") is not None:  # fused Winograd block order: 0 ko block fastest, 1 patch fastest, -1 rule
        L.so_wino_fused_kb_major(int(os.environ["WINO_KBMAJOR"]))
    if os.environ.get("WINO_KB32") is not None:     # 1 (default): 32 output channels per block; 0: 64 where Ko >= 64
        L.so_wino_fused_force_kb32(int(os.environ["WINO_KB32"]))
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    ws = ops.workspace(dev)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    x = torch.randn(n * h * w, ci, device=dev)
    wt = torch.randn(co, k * k * ci, device=dev) * 0.05
    y = torch.randn(n * ho * wo, co, device=dev)
    dx = torch.empty_like(x)
    dw = torch.zeros_like(wt)
    if mode == "wino":
        u = torch.empty(L.so_wino_fused_weight_floats(co, ci, 0), device=dev)
        L.so_wino_fused_weights(wt.data_ptr(), u.data_ptr(), co, co, ci, 0, st)
    calls = {
        "fprop": lambda: L.so_conv2d_fprop(x.data_ptr(), ci, wt.data_ptr(), None, y.data_ptr(), co, n, h, w, ci, co, k, k, s, p, 0, 0.0,
                                           ws.data_ptr(), ws.numel() * 4, st),
        "dgrad": lambda: L.so_conv2d_dgrad(y.data_ptr(), co, wt.data_ptr(), dx.data_ptr(), ci, n, h, w, ci, co, k, k, s, p,
                                           ws.data_ptr(), ws.numel() * 4, st),
        "wgrad": lambda: L.so_conv2d_wgrad_acc(y.data_ptr(), co, x.data_ptr(), ci, dw.data_ptr(), n, h, w, ci, co, k, k, s, p,
                                               ws.data_ptr(), ws.numel() * 4, st),
        "wino": lambda: L.so_wino_fused_conv3x3(x.data_ptr(), ci, u.data_ptr(), None, 0, None, y.data_ptr(), co, n, h, w, ci, co, 1, 0.0, st),
    }
    fn = calls[mode]
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    gf = 2.0 * n * ho * wo * co * k * k * ci / 1e9
    print(f"{mode} N={n} {h}x{w} {ci}->{co} k{k}s{s}p{p}: {us:.1f} us/launch, {gf / us * 1e3:.1f} TFLOP/s (direct-convolution FLOPs)")


if __name__ == "__main__":
    main()

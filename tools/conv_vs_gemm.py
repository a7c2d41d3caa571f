"""Same (M, N, K) through the engine as a 3x3 convolution, as a 1x1 convolution and as a plain GEMM (forced 64x64 and
128x128 tiles, warmed-up chip): isolates what the implicit-GEMM gather / decode costs against the GEMM core."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shineon_virtual_tryon_amd as pkg  # noqa: E402
from shineon_virtual_tryon_amd import ops  # noqa: E402

L = pkg.lib()
dev = torch.device("cuda", 0)
ws = ops.workspace(dev)
st = torch.cuda.current_stream().cuda_stream


def bench(fn, flops):
    for _ in range(int(60e-3 / (flops / 100e12)) + 2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return flops / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12


for (n, h, w, c, ko) in ((8, 64, 48, 256, 256), (8, 32, 24, 512, 512), (8, 256, 192, 64, 64), (4, 128, 96, 256, 64)):
    M, K = n * h * w, 9 * c
    x = torch.randn(M, c, device=dev)
    w3 = torch.randn(ko, K, device=dev) * 0.05
    y = torch.empty(M, ko, device=dev)
    xg = torch.randn(M, K, device=dev)
    flops = 2.0 * M * ko * K
    for tile in ((64, 64, 1), (128, 128, 1), (128, 128, 2), (64, 128, 1)):
        L.so_igemm_force(*tile)
        r3 = bench(lambda: L.so_conv2d_fprop(x.data_ptr(), c, w3.data_ptr(), None, y.data_ptr(), ko, n, h, w, c, ko, 3, 3, 1, 1, 0, 0.0,
                                             ws.data_ptr(), ws.numel() * 4, st), flops)
        r1 = bench(lambda: L.so_conv2d_fprop(xg.data_ptr(), K, w3.data_ptr(), None, y.data_ptr(), ko, n, h, w, K, ko, 1, 1, 1, 0, 0, 0.0,
                                             ws.data_ptr(), ws.numel() * 4, st), flops)
        rg = bench(lambda: L.so_gemm_batched(0, 1, M, ko, K, xg.data_ptr(), K, 0, w3.data_ptr(), K, 0, y.data_ptr(), ko, 0, 1, None, None,
                                             None, 0, 0, 0, 0.0, ws.data_ptr(), ws.numel() * 4, st), flops)
        print(f"M={M} N={ko} K={K} tile {tile}: conv3x3 {r3:6.1f}  conv1x1 {r1:6.1f}  gemm {rg:6.1f} TFLOP/s", flush=True)
    L.so_igemm_force(0, 0, 0)

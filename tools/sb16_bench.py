"""Split-bf16 3x3 convolution (csrc/sb16.hip) vs the fp32 engine on the VGG layer shapes (2B = 8 images forward)."""
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
    for _ in range(int(60e-3 / (flops / 150e12)) + 2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    return us, flops / (us * 1e-6) / 1e12


for (n, h, w, c, ko) in ((8, 128, 96, 64, 128), (8, 128, 96, 128, 128), (8, 64, 48, 128, 256), (8, 64, 48, 256, 256), (8, 32, 24, 256, 512),
                         (8, 32, 24, 512, 512), (8, 16, 12, 512, 512), (4, 64, 48, 256, 256), (4, 128, 96, 128, 64)):
    M, K = n * h * w, 9 * c
    x = torch.randn(M, c, device=dev)
    wt = torch.randn(ko, K, device=dev) * 0.05
    bias = torch.randn(ko, device=dev)
    y = torch.empty(M, ko, device=dev)
    xh, xm = ops._sb16_planes(x.data_ptr(), c, c, M, dev)
    wh = torch.empty(ko * K, dtype=torch.bfloat16, device=dev)
    wm = torch.empty_like(wh)
    L.so_sb16_prep_weights(wt.data_ptr(), ko, c, c, 0, wh.data_ptr(), wm.data_ptr(), st)
    yh = torch.empty(M, ko, dtype=torch.bfloat16, device=dev)
    ym = torch.empty_like(yh)
    flops = 2.0 * M * ko * K
    t32 = bench(lambda: L.so_conv2d_fprop(x.data_ptr(), c, wt.data_ptr(), bias.data_ptr(), y.data_ptr(), ko, n, h, w, c, ko, 3, 3, 1, 1, 1, 0.0,
                                          ws.data_ptr(), ws.numel() * 4, st), flops)
    ts = bench(lambda: L.so_sb16_conv3x3(xh.data_ptr(), xm.data_ptr(), wh.data_ptr(), wm.data_ptr(), bias.data_ptr(), None, y.data_ptr(), ko,
                                         None, None, n, h, w, c, ko, 1, st), flops)
    te = bench(lambda: L.so_sb16_conv3x3(xh.data_ptr(), xm.data_ptr(), wh.data_ptr(), wm.data_ptr(), bias.data_ptr(), None, y.data_ptr(), ko,
                                         yh.data_ptr(), ym.data_ptr(), n, h, w, c, ko, 1, st), flops)
    tsp = bench(lambda: L.so_sb16_split(x.data_ptr(), c, c, xh.data_ptr(), xm.data_ptr(), M, st), flops)
    print(f"N={n} {h}x{w} C={c} Ko={ko}: fp32 engine {t32[0]:7.1f} us ({t32[1]:5.1f} TF) | sb16 {ts[0]:7.1f} us ({ts[1]:5.1f} TF) | "
          f"sb16 + planes out {te[0]:7.1f} us | split pass of the input {tsp[0]:6.1f} us", flush=True)

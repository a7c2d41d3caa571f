"""Times a few layer-sized igemm launches with the library named by SHINEON_LIB (tools/ablate_igemm.sh variants).

    SHINEON_LIB=.../libshineon_hip_abl2.so python tools/ablate_bench.py TAG
"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SHINEON_PLANS", "none")
import shineon_virtual_tryon_amd as pkg  # noqa: E402
from shineon_virtual_tryon_amd import ops  # noqa: E402

# (name, mode, N, H, W, Cin, Cout, k, s, p, (bm, bn, splitk))
CASES = [
    ("gmm3x3 512>512@16x12", "wgrad", 4, 16, 12, 512, 512, 3, 1, 1, (64, 64, 1)),
    ("4x4s2 256>512@32x24", "wgrad", 4, 32, 24, 256, 512, 4, 2, 1, (64, 64, 1)),
    ("vgg3_x 256>256@64x48 x8", "fprop", 8, 64, 48, 256, 256, 3, 1, 1, (64, 64, 1)),
    ("vgg3_x 256>256@64x48 x8", "fprop", 8, 64, 48, 256, 256, 3, 1, 1, (128, 128, 1)),
    ("4x4s2 64>128@128x96", "fprop", 4, 128, 96, 64, 128, 4, 2, 1, (64, 64, 1)),
    ("4x4s2 64>128@128x96", "dgrad", 4, 128, 96, 64, 128, 4, 2, 1, (64, 64, 1)),
    ("4x4s2 128>256@64x48", "dgrad", 4, 64, 48, 128, 256, 4, 2, 1, (128, 64, 1)),
    ("3x3 512>128@64x48", "wgrad", 4, 64, 48, 512, 128, 3, 1, 1, (128, 128, 8)),
]


# Winograd-domain batched GEMMs (csrc/wino.hip): (M tiles, N = Ko, K = C, batch = transform points), forced tile
GEMMS = [(1536, 256, 256, 36, (64, 64, 1)), (1536, 256, 256, 36, (128, 128, 1)), (384, 512, 512, 36, (64, 64, 1)),
         (192, 512, 512, 36, (64, 64, 1)), (3072, 128, 512, 16, (64, 64, 1)), (3072, 128, 512, 16, (128, 64, 1)),
         (192, 512, 192, 4, (64, 64, 1))]


def timed(fn, L):
    ts = []
    for rnd in range(6):
        assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    return statistics.median(ts)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "base"
    L = pkg.lib()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    ws = ops.workspace(dev)
    for M, N, K, nb, (bm, bn, sk) in GEMMS:
        A = torch.randn(nb, M, K, device=dev)
        B = torch.randn(nb, N, K, device=dev)
        C = torch.empty(nb, M, N, device=dev)
        L.so_igemm_force(bm, bn, sk)
        us = timed(lambda: L.so_gemm_batched(0, 1, M, N, K, A.data_ptr(), K, M * K, B.data_ptr(), K, N * K, C.data_ptr(), N, M * N, nb,
                                             None, None, None, 0, 0, 0, 0.0, ws.data_ptr(), ws.numel() * 4, st), L)
        L.so_igemm_force(0, 0, 0)
        print(f"{tag:6s} gemm {M}x{N}x{K} b{nb:<3d}          nt {bm}x{bn} sk{sk}: {us:7.1f} us  {2.0 * M * N * K * nb / us / 1e6:6.1f} TF", flush=True)
    for name, mode, n, h, w, ci, co, k, s, p, (bm, bn, sk) in CASES:
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        x = torch.randn(n * h * w, ci, device=dev)
        wt = torch.randn(co, k * k * ci, device=dev) * 0.05
        y = torch.randn(n * ho * wo, co, device=dev)
        dx = torch.empty_like(x)
        dw = torch.zeros_like(wt)
        fn = {
            "fprop": lambda: L.so_conv2d_fprop(x.data_ptr(), ci, wt.data_ptr(), None, y.data_ptr(), co, n, h, w, ci, co, k, k, s, p, 0,
                                               0.0, ws.data_ptr(), ws.numel() * 4, st),
            "dgrad": lambda: L.so_conv2d_dgrad(y.data_ptr(), co, wt.data_ptr(), dx.data_ptr(), ci, n, h, w, ci, co, k, k, s, p,
                                               ws.data_ptr(), ws.numel() * 4, st),
            "wgrad": lambda: L.so_conv2d_wgrad_acc(y.data_ptr(), co, x.data_ptr(), ci, dw.data_ptr(), n, h, w, ci, co, k, k, s, p,
                                                   ws.data_ptr(), ws.numel() * 4, st),
        }[mode]
        L.so_igemm_force(bm, bn, sk)
        ts = []
        for rnd in range(6):
            assert fn() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        L.so_igemm_force(0, 0, 0)
        gf = 2.0 * n * ho * wo * co * k * k * ci / 1e9
        us = statistics.median(ts)
        print(f"{tag:6s} {name:26s} {mode} {bm}x{bn} sk{sk}: {us:7.1f} us  {gf / us * 1e3:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()

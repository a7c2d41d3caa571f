"""Micro-benchmark of the fp32 MFMA implicit-GEMM kernels on the hot path's layer shapes (bs=4).

    python tools/igemm_bench.py [--modes fprop,dgrad,wgrad] [--filter vgg] [--csv out.csv]

For every (layer, mode) it times the auto-planned launch and every forced tile / split-K variant and prints
TFLOP/s (algorithmic FLOPs / kernel time incl. the split-K reduce).  Used to calibrate the tile planner.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shineon_virtual_tryon_amd as pkg  # noqa: E402
from shineon_virtual_tryon_amd import ops  # noqa: E402

# (name, N, H, W, Cin, Cout, k, stride, pad)   H, W = INPUT size
LAYERS = [
    ("vgg1_2", 4, 256, 192, 64, 64, 3, 1, 1),
    ("vgg2_1", 4, 128, 96, 64, 128, 3, 1, 1),
    ("vgg2_2", 4, 128, 96, 128, 128, 3, 1, 1),
    ("vgg3_1", 4, 64, 48, 128, 256, 3, 1, 1),
    ("vgg3_x", 4, 64, 48, 256, 256, 3, 1, 1),
    ("vgg4_1", 4, 32, 24, 256, 512, 3, 1, 1),
    ("vgg4_x", 4, 32, 24, 512, 512, 3, 1, 1),
    ("vgg5_1", 4, 16, 12, 512, 512, 3, 1, 1),
    ("unet_d0", 4, 256, 192, 12, 64, 4, 2, 1),
    ("unet_d1", 4, 128, 96, 64, 128, 4, 2, 1),
    ("unet_d2", 4, 64, 48, 128, 256, 4, 2, 1),
    ("unet_d3", 4, 32, 24, 256, 512, 4, 2, 1),
    ("unet_d4", 4, 16, 12, 512, 512, 4, 2, 1),
    ("unet_d5", 4, 8, 6, 512, 512, 4, 2, 1),
    ("unet_u5", 4, 8, 6, 512, 512, 3, 1, 1),
    ("unet_u4", 4, 16, 12, 1024, 512, 3, 1, 1),
    ("unet_u3", 4, 32, 24, 1024, 256, 3, 1, 1),
    ("unet_u2", 4, 64, 48, 512, 128, 3, 1, 1),
    ("unet_u1", 4, 128, 96, 256, 64, 3, 1, 1),
    ("unet_u0", 4, 256, 192, 128, 4, 3, 1, 1),
    ("gmm_a0", 4, 256, 192, 24, 64, 4, 2, 1),
    ("gmm_3x3", 4, 16, 12, 512, 512, 3, 1, 1),
    ("gmm_r0", 4, 16, 12, 192, 512, 4, 2, 1),
]
VARIANTS = [(0, 0, 0), (64, 64, 1), (128, 64, 1), (64, 128, 1), (128, 128, 1), (64, 64, 2), (64, 64, 4), (64, 64, 8),
            (128, 128, 2), (128, 128, 4), (128, 64, 4), (64, 128, 4), (64, 64, 16), (128, 128, 16), (64, 128, 32)]


def time_call(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="fprop,dgrad,wgrad")
    ap.add_argument("--filter", default="")
    ap.add_argument("--csv", default="")
    args = ap.parse_args()
    L = pkg.lib()
    dev = torch.device("cuda", 0)
    ws = ops.workspace(dev)
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for name, n, h, w, ci, co, k, s, p in LAYERS:
        if args.filter and args.filter not in name:
            continue
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        x = torch.randn(n * h * w, ci, device=dev)
        wt = torch.randn(co, k * k * ci, device=dev) * 0.05
        y = torch.randn(n * ho * wo, co, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(wt)
        flops = 2.0 * n * ho * wo * co * k * k * ci
        calls = {
            "fprop": lambda: L.so_conv2d_fprop(x.data_ptr(), ci, wt.data_ptr(), None, y.data_ptr(), co, n, h, w, ci, co, k, k, s, p,
                                               0, 0.0, ws.data_ptr(), ws.numel() * 4, st),
            "dgrad": lambda: L.so_conv2d_dgrad(y.data_ptr(), co, wt.data_ptr(), dx.data_ptr(), ci, n, h, w, ci, co, k, k, s, p,
                                               ws.data_ptr(), ws.numel() * 4, st),
            "wgrad": lambda: L.so_conv2d_wgrad(y.data_ptr(), co, x.data_ptr(), ci, dw.data_ptr(), n, h, w, ci, co, k, k, s, p,
                                               ws.data_ptr(), ws.numel() * 4, st),
        }
        for mode in args.modes.split(","):
            if mode == "dgrad" and (co % 4 or ci % 4):
                continue
            res = {}
            for v in VARIANTS:
                L.so_igemm_force(*v)
                err = calls[mode]()
                if err != 0:
                    continue
                res[v] = flops / time_call(calls[mode]) / 1e12
            L.so_igemm_force(0, 0, 0)
            best = max((t, v) for v, t in res.items() if v != (0, 0, 0))
            line = f"{name:8s} {mode:5s} GF={flops / 1e9:7.2f} auto={res[(0, 0, 0)]:6.1f} best={best[0]:6.1f}@{best[1]} | " + \
                   " ".join(f"{v[0]}x{v[1]}/{v[2]}:{t:.0f}" for v, t in res.items() if v != (0, 0, 0))
            print(line, flush=True)
            rows.append((name, mode, flops, res))
    if args.csv:
        with open(args.csv, "w") as f:
            f.write("layer,mode,gflop," + ",".join(f"{v[0]}x{v[1]}/{v[2]}" for v in VARIANTS) + "\n")
            for name, mode, flops, res in rows:
                f.write(f"{name},{mode},{flops / 1e9:.3f}," + ",".join(f"{res.get(v, float('nan')):.2f}" for v in VARIANTS) + "\n")


if __name__ == "__main__":
    main()

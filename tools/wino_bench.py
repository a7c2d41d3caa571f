"""Direct implicit-GEMM convolution vs Winograd F(2x2,3x3) (csrc/wino.hip) on the VGG19 layer shapes of the perceptual loss
(x || y = 8 images forward, 4 images input gradient at bs = 4).  Prints time, direct-equivalent TFLOP/s and max error against an
fp64 convolution of the same operands (first case only).  GPU box:  python tools/wino_bench.py [--batch 8]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shineon_virtual_tryon_amd as pkg  # noqa: E402
from shineon_virtual_tryon_amd import ops  # noqa: E402

LAYERS = [(256, 192, 64, 64), (128, 96, 64, 128), (128, 96, 128, 128), (64, 48, 128, 256), (64, 48, 256, 256),
          (32, 24, 256, 512), (32, 24, 512, 512), (16, 12, 512, 512)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[8, 4])
    ap.add_argument("--layers", type=int, nargs="*", default=None)
    args = ap.parse_args()
    L = pkg.lib()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    ws = ops.workspace(dev)
    print("batch,H,W,C,Ko,direct_us,direct_tflops,wino_us,wino_eq_tflops,speedup,in_us,gemm_us,out_us,max_err_direct,max_err_wino")
    for nb in args.batch:
        for li, (h, w, c, ko) in enumerate(LAYERS):
            if args.layers and li not in args.layers:
                continue
            torch.manual_seed(li)
            x = torch.randn(nb * h * w, c, device=dev)
            wt = torch.randn(ko, 3, 3, c, device=dev) * (2.0 / (9 * c)) ** 0.5
            bias = torch.randn(ko, device=dev) * 0.1
            y0 = torch.empty(nb * h * w, ko, device=dev)
            y1 = torch.empty_like(y0)
            U = torch.empty(16, ko, c, device=dev)
            assert L.so_wino_weights(wt.data_ptr(), U.data_ptr(), ko, ko, c, 0, st) == 0
            wws = torch.empty(L.so_wino_ws_floats(nb, h, w, c, ko), device=dev)
            direct = lambda: L.so_conv2d_fprop(x.data_ptr(), c, wt.data_ptr(), bias.data_ptr(), y0.data_ptr(), ko, nb, h, w, c, ko, 3, 3,  # noqa: E731
                                               1, 1, 1, 0.0, ws.data_ptr(), ws.numel() * 4, st)
            wino = lambda: L.so_wino_conv3x3(x.data_ptr(), c, U.data_ptr(), bias.data_ptr(), ko, None, y1.data_ptr(), ko, nb, h, w, c,  # noqa: E731
                                             ko, 1, 0.0, wws.data_ptr(), wws.numel() * 4, ws.data_ptr(), ws.numel() * 4, st)
            Uf = torch.empty(L.so_wino_fused_weight_floats(ko, c, 0), device=dev)
            assert L.so_wino_fused_weights(wt.data_ptr(), Uf.data_ptr(), ko, ko, c, 0, st) == 0
            y2 = torch.empty_like(y0)
            fused = lambda: L.so_wino_fused_conv3x3(x.data_ptr(), c, Uf.data_ptr(), bias.data_ptr(), ko, None, y2.data_ptr(), ko, nb, h, w,  # noqa: E731
                                                    c, ko, 1, 0.0, st)
            assert direct() == 0 and wino() == 0 and fused() == 0
            torch.cuda.synchronize()
            T = nb * (h // 2) * (w // 2)
            V, Mx = wws[:16 * T * c], wws[16 * T * c:]
            t_gemm = timeit(lambda: L.so_gemm_batched(0, 1, T, ko, c, V.data_ptr(), c, T * c, U.data_ptr(), c, ko * c, Mx.data_ptr(), ko,
                                                      T * ko, 16, None, None, None, 0, 0, 0, 0.0, ws.data_ptr(), ws.numel() * 4, st))
            td, tw, tf = timeit(direct), timeit(wino), timeit(fused)
            err_f = float((y2 - y0).abs().max())
            U4 = torch.empty(36, ko, c, device=dev)
            assert L.so_wino4_weights(wt.data_ptr(), U4.data_ptr(), ko, ko, c, 0, st) == 0
            wws4 = torch.empty(L.so_wino4_ws_floats(nb, h, w, c, ko), device=dev)
            y4 = torch.empty_like(y0)
            f44 = lambda: L.so_wino4_conv3x3(x.data_ptr(), c, U4.data_ptr(), bias.data_ptr(), ko, None, y4.data_ptr(), ko, nb, h, w, c,  # noqa: E731
                                             ko, 1, 0.0, wws4.data_ptr(), wws4.numel() * 4, ws.data_ptr(), ws.numel() * 4, st)
            assert f44() == 0
            t44 = timeit(f44)
            err_44 = float((y4 - y0).abs().max())
            L.so_wino_fused_force_kb32(0)   # the 64-output-channel / two-blocks-per-CU instantiation for comparison
            tf32 = timeit(fused)
            L.so_wino_fused_force_kb32(-1)

            err_d = err_w = float("nan")
            if nb * h * w <= 8 * 64 * 48:
                ref = torch.nn.functional.conv2d(x.view(nb, h, w, c).permute(0, 3, 1, 2).double().cpu(), wt.permute(0, 3, 1, 2).double().cpu(),
                                                 bias.double().cpu(), padding=1).relu().permute(0, 2, 3, 1).reshape(-1, ko)
                err_d = float((y0.cpu().double() - ref).abs().max())
                err_w = float((y1.cpu().double() - ref).abs().max())
            else:
                err_w = float((y1 - y0).abs().max())
            gf = 2.0 * nb * h * w * ko * 9 * c / 1e9
            print(f"{nb},{h},{w},{c},{ko},{td:.1f},{gf / td * 1e3:.1f},{tw:.1f},{gf / tw * 1e3:.1f},{td / tw:.2f},-,{t_gemm:.1f},-,"
                  f"{err_d:.2e},{err_w:.2e},fused_us={tf:.1f},fused_eq_tflops={gf / tf * 1e3:.1f},fused_speedup={td / tf:.2f},"
                  f"fused_vs_direct_maxdiff={err_f:.2e},fused_kb64_us={tf32:.1f},f44_us={t44:.1f},f44_vs_direct_maxdiff={err_44:.2e}", flush=True)


if __name__ == "__main__":
    main()

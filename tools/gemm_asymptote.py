"""Asymptotic throughput of the fp32 MFMA GEMM core (no conv gather) at large square sizes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shineon_virtual_tryon_amd as pkg
from shineon_virtual_tryon_amd import ops
L = pkg.lib(); dev = torch.device("cuda", 0); ws = ops.workspace(dev); st = torch.cuda.current_stream().cuda_stream
for n in (2048, 4096):
    a = torch.randn(n, n, device=dev); b = torch.randn(n, n, device=dev); c = torch.empty(n, n, device=dev)
    for ta, tb in ((0, 1), (0, 0), (1, 0)):
        for tile in ((64, 64), (128, 64), (64, 128), (128, 128)):
            L.so_igemm_force(tile[0], tile[1], 1)
            f = lambda: L.so_gemm_batched(ta, tb, n, n, n, a.data_ptr(), n, 0, b.data_ptr(), n, 0, c.data_ptr(), n, 0, 1, None, None, None, 0, 0, 0, 0.0, ws.data_ptr(), ws.numel() * 4, st)
            assert f() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 5 * 1e-3
            print(f"n={n} ta={ta} tb={tb} tile={tile}: {2.0 * n ** 3 / t / 1e12:6.1f} TFLOP/s", flush=True)
L.so_igemm_force(0, 0, 0)

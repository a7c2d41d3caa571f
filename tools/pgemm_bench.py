"""Persistent Winograd-domain GEMM (csrc/pgemm.hip) vs the general engine with its committed plans, on the c4 step's shapes."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shineon_virtual_tryon_amd as pkg  # noqa: E402

SHAPES = [(1536, 256, 256, 36), (384, 512, 512, 36), (768, 256, 256, 36), (192, 512, 512, 36), (3072, 128, 512, 16),
          (3072, 128, 256, 16), (192, 1024, 256, 36), (384, 512, 256, 36), (192, 256, 1024, 36), (96, 512, 512, 36),
          (1536, 256, 512, 36), (6144, 128, 256, 16)]


def timed(fn):
    ts = []
    for rnd in range(6):
        assert fn() in (0, 1)
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
    L = pkg.lib()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(256 << 20, device=dev)
    for M, N, K, nb in SHAPES:
        A = torch.randn(nb, M, K, device=dev)
        B = torch.randn(nb, N, K, device=dev)
        C = torch.empty(nb, M, N, device=dev)
        eng = lambda: L.so_gemm_batched(0, 1, M, N, K, A.data_ptr(), K, M * K, B.data_ptr(), K, N * K, C.data_ptr(), N, M * N, nb, None,  # noqa: E731
                                        None, None, 0, 0, 0, 0.0, ws.data_ptr(), ws.numel() * 4, st)
        per = lambda: L.so_pgemm_nt(M, N, K, A.data_ptr(), K, M * K, B.data_ptr(), K, N * K, C.data_ptr(), N, M * N, nb, st)  # noqa: E731
        te = timed(eng)
        applicable = per() == 0   # 0 = launched, SO_NOT_APPLICABLE (-3) = declined
        tp = timed(per) if applicable else float("nan")
        gf = 2.0 * M * N * K * nb / 1e9
        print(f"{M}x{N}x{K} b{nb}: engine {te:6.1f} us {gf / te * 1e3:6.1f} TF   persistent {tp:6.1f} us {gf / tp * 1e3 if applicable else 0:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()

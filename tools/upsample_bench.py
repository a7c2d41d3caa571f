"""Bilinear x2 up-sampling (+ fused activation, unmaterialised concat) on the U-Net up-path levels at bs = 4: time and HBM
rate of the forward and backward kernels.  Run twice - SO_UPSAMPLE_TILED=1 (default) and =0 - to compare the LDS-tiled
kernels of the large levels with the element-per-thread form (csrc/elementwise.hip)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shineon_virtual_tryon_amd import ops  # noqa: E402
from shineon_virtual_tryon_amd._lib import lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    n = 4
    print(f"SO_UPSAMPLE_TILED={os.environ.get('SO_UPSAMPLE_TILED', '1')}")
    print("level,C1,C2,act,fwd_us,fwd_GBs,bwd_us,bwd_GBs")
    for (h, w, c1, c2) in [(128, 96, 64, 64), (64, 48, 128, 128), (32, 24, 256, 256), (16, 12, 512, 512), (128, 96, 128, 0)]:
        for act in (ops.ACT_CODES["gelu"], ops.ACT_CODES[None]):
            c = c1 + c2
            a = torch.randn(n, h, w, c1, device=dev)
            b = torch.randn(n, h, w, c2, device=dev) if c2 else None
            y = torch.empty(n, 2 * h, 2 * w, c, device=dev)
            dy = torch.randn_like(y)
            da = torch.empty_like(a)
            db = torch.empty_like(b) if c2 else None
            bp = lambda t: t.data_ptr() if t is not None else None

            def fwd():
                return L.so_upsample2x_cat_fwd(a.data_ptr(), c1, c1, bp(b), c2, c2, y.data_ptr(), c, n, h, w, act, 0.0, st)

            def bwd():
                return L.so_upsample2x_cat_bwd(a.data_ptr(), c1, c1, bp(b), c2, c2, dy.data_ptr(), c, da.data_ptr(), c1, bp(db), c2,
                                               n, h, w, act, 0.0, st)

            res = []
            for fn, nbytes in ((fwd, (a.numel() + (b.numel() if c2 else 0) + y.numel()) * 4),
                               (bwd, (dy.numel() + (2 if act else 1) * (a.numel() + (b.numel() if c2 else 0))) * 4)):
                for _ in range(5):
                    assert fn() == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 50
                res += [f"{us:.1f}", f"{nbytes / us / 1e3:.0f}"]
            print(f"{h}x{w},{c1},{c2},{'gelu' if act else 'none'}," + ",".join(res))


if __name__ == "__main__":
    main()

"""What one all-reduce costs on a ONE-rank RCCL group (no wire): host time per call, device time on the calling stream
(event pair around all_reduce(async_op=True).wait()), for a small and a slab-sized message.  Explains the exchange overhead
that bench.py reports under SHINEON_SINGLE_RANK_GROUP=1."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SHINEON_SINGLE_RANK_GROUP"] = "1"
from shineon_virtual_tryon_amd import trainer  # noqa: E402

trainer.init_distributed()
dev = torch.device("cuda", 0)
filler = torch.randn(64 << 20, device=dev)
for n in (256, 22 << 20):
    t = torch.randn(n, device=dev)
    for mode in ("idle", "busy"):
        for _ in range(3):
            dist.all_reduce(t, async_op=True).wait()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        if mode == "busy":   # the stream has work queued: the host runs ahead, only device-side cost shows
            for _ in range(20):
                filler.mul_(1.0)
        e0.record()
        h0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(t, async_op=True).wait()
        h1 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        print(f"{n * 4 / 1e6:9.3f} MB {mode}: host {1e6 * (h1 - h0) / reps:7.1f} us/call, stream {1e3 * e0.elapsed_time(e1) / reps:7.1f} us/call"
              + (" (incl. 20 filler kernels / reps)" if mode == "busy" else ""), flush=True)
dist.destroy_process_group()

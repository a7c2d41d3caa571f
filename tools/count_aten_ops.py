"""Lists the ATen (non-HIP-library) GPU ops that still run inside one eager training step of each model, with input
shapes - the candidates for fusion into the library's own kernels.   python tools/count_aten_ops.py  (GPU box)"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from shineon_virtual_tryon_amd.data import synthetic_batch  # noqa: E402
from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel  # noqa: E402
from shineon_virtual_tryon_amd.warp_model import WarpModel  # noqa: E402

dev = torch.device("cuda", 0)
warp = WarpModel(bench.hparams(person_inputs=["agnostic", "cocopose"])).to(dev).train()
unet = UnetMaskModel(bench.hparams(person_inputs=["agnostic", "densepose"])).to(dev).train()
warp.global_step = unet.global_step = 1
(optw,), _ = warp.configure_optimizers()
(optu,), _ = unet.configure_optimizers()
batch = synthetic_batch(4, dev, seed=420)


def step(model, opt, b):
    opt.zero_grad()
    res = model.training_step(b, 0)
    res.minimize.backward()
    opt.step()


for name, model, opt in (("warp", warp, optw), ("unet", unet, optu)):
    b = dict(batch)
    if name == "unet":
        b["cloth"] = warp.warped_cloth.detach()
    step(model, opt, b)
    step(model, opt, b)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step(model, opt, b)
        torch.cuda.synchronize()
    print(f"== {name}: aten ops with self GPU time")
    rows = [e for e in prof.key_averages(group_by_input_shape=True)
            if e.key.startswith("aten::") and getattr(e, "self_device_time_total", 0) > 0]
    for e in sorted(rows, key=lambda e: -e.self_device_time_total):
        print(f"{e.count:4d} {e.key:28s} {e.self_device_time_total:9.1f} us  {str(e.input_shapes)[:110]}")

"""GPU kernel time of one eager training step of each model, by kernel (torch profiler).   python tools/model_kernel_times.py"""
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
    for _ in range(3):
        step(model, opt, b)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step(model, opt, b)
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
    tot = sum(e.self_device_time_total for e in rows)
    print(f"== {name}: {tot / 3e3:.3f} ms of kernel time per step, {sum(e.count for e in rows) / 3:.0f} kernels")
    for e in rows[:int(os.environ.get("TOPK", "22"))]:
        print(f"{e.count / 3:6.1f} x {e.self_device_time_total / e.count:8.1f} us = {e.self_device_time_total / 3e3:7.3f} ms  {e.key[:90]}")

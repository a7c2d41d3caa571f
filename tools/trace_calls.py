"""Counts, per Python call site in ops.py / the models, the launches of one C-ABI entry point during a training step.
    python tools/trace_calls.py so_copy2d        (GPU box)"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import shineon_virtual_tryon_amd as pkg  # noqa: E402
from shineon_virtual_tryon_amd.data import synthetic_batch  # noqa: E402
from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel  # noqa: E402
from shineon_virtual_tryon_amd.warp_model import WarpModel  # noqa: E402

name = sys.argv[1]
L = pkg.lib()
orig = getattr(L, name)
sites = collections.Counter()
on = [False]


def wrapper(*a):
    if on[0]:
        st = [f for f in traceback.extract_stack()[:-1] if "shineon-virtual-tryon_amd" in f.filename]
        sites[" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:][::-1])] += 1
    return orig(*a)


setattr(L, name, wrapper)
dev = torch.device("cuda", 0)
warp = WarpModel(bench.hparams(person_inputs=["agnostic", "cocopose"])).to(dev).train()
unet = UnetMaskModel(bench.hparams(person_inputs=["agnostic", "densepose"])).to(dev).train()
warp.global_step = unet.global_step = 1
(optw,), _ = warp.configure_optimizers()
(optu,), _ = unet.configure_optimizers()
batch = synthetic_batch(4, dev, seed=420)
for it in range(2):
    on[0] = it == 1
    for model, opt in ((warp, optw), (unet, optu)):
        b = dict(batch)
        if model is unet:
            b["cloth"] = warp.warped_cloth.detach()
        opt.zero_grad()
        res = model.training_step(b, 0)
        res.minimize.backward()
        opt.step()
torch.cuda.synchronize()
for k, v in sites.most_common():
    print(f"{v:4d}  {k}")
print("total", sum(sites.values()))

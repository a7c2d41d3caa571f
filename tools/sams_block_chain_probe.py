"""Diagnostic: [resize 0.5] -> block A (identity shortcut) -> block B -> sum, gradients of block A against the oracle."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sams_oracle as so  # noqa: E402
from oracle.procedural import procedural_state_dict, shapes_of  # noqa: E402

import shineon_virtual_tryon_amd  # noqa: E402,F401
from shineon_virtual_tryon_amd import ops_sams  # noqa: E402
from shineon_virtual_tryon_amd.networks.sams import AnySpadeResBlock, MultiSpade  # noqa: E402

norm_G = sys.argv[1] if len(sys.argv) > 1 else "spadeinstance3x3"
labels = {"agnostic": 4, "cloth": 3, "flow": 2}
A = AnySpadeResBlock(32, 32, norm_G, labels, MultiSpade, "relu")
B = AnySpadeResBlock(32, 32, norm_G, labels, MultiSpade, "relu")
sd = procedural_state_dict({**{"a." + k: v for k, v in shapes_of(A.state_dict()).items()},
                            **{"b." + k: v for k, v in shapes_of(B.state_dict()).items()}})
A.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith("a.")})
B.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith("b.")})
A, B = A.cuda().train(), B.cuda().train()
torch.manual_seed(7)
x0 = torch.randn(2, 32, 32, 24)
seg = {k: torch.randn(2, c, 64, 48) for k, c in labels.items()}
gout = torch.randn(2, 32, 16, 12)
hp = argparse.Namespace(norm_G=norm_G, activation="relu")
for mode in ("leaf", "resized", "chain"):
    osd = {k: v.double().clone().requires_grad_(v.is_floating_point() and not k.endswith(("running_mean", "running_var", "weight_u", "weight_v", "num_batches_tracked"))) if v.is_floating_point() else v.clone() for k, v in sd.items()}
    xin = x0.double().clone().requires_grad_(True)
    oseg = {k: v.double() for k, v in seg.items()}
    x = F.interpolate(xin, scale_factor=0.5, mode="nearest") if mode != "leaf" else xin[:, :, ::2, ::2]
    y = so.spade_resblock(osd, "a", x, oseg, hp, True)
    if mode == "chain":
        y = so.spade_resblock(osd, "b", y, oseg, hp, True)
    y.backward(gout.double())
    for m in (A, B):
        m.zero_grad(set_to_none=True)
    dx0 = x0.clone().cuda().requires_grad_(True)
    dseg = {k: v.cuda() for k, v in seg.items()}
    if mode == "leaf":
        xl = x0[:, :, ::2, ::2].contiguous().cuda().requires_grad_(True)
        yy = A(xl, dseg)
    else:
        yy = A(ops_sams.resize_nearest(dx0, scale_factor=0.5), dseg)
    if mode == "chain":
        yy = B(yy, dseg)
    yy.backward(gout.cuda())
    worst = ("", 0.0)
    for k, p in A.named_parameters():
        ref = osd["a." + k].grad
        if p.grad is None or ref is None or ref.abs().max() < 1e-9:
            continue
        e = (p.grad.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        if e > worst[1]:
            worst = (k, e)
    gx = (dx0.grad if mode != "leaf" else xl.grad)
    refx = xin.grad if mode != "leaf" else xin.grad[:, :, ::2, ::2]
    print(mode, "worst param grad", worst, "input grad", (gx.double().cpu() - refx).abs().max().item() / refx.abs().max().item())

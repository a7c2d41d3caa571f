"""Diagnostic: every _Conv2dFn.backward of the generator step re-checked against torch.nn.grad on the CPU (fp64)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sams_helpers as sh  # noqa: E402
from oracle.procedural import procedural_state_dict  # noqa: E402

import shineon_virtual_tryon_amd  # noqa: E402,F401
from shineon_virtual_tryon_amd import ops  # noqa: E402
from shineon_virtual_tryon_amd.data import synthetic_batch  # noqa: E402
from shineon_virtual_tryon_amd.sams_model import SamsModel  # noqa: E402

tag = "progressive"
g = sh.load_golden(tag)
hp = sh.sams_hparams(**sh.SAMS_VARIANTS[tag])
sd = procedural_state_dict(sh.golden_shapes(g))
batch = synthetic_batch(2, "cpu", height=hp.fine_height, width=hp.fine_width, n_frames=hp.n_frames_total, smooth=True)
orig = ops._Conv2dFn.backward
REC = []


def traced(ctx, dy):
    out = orig(ctx, dy)
    xr, w, y = ctx.saved_tensors
    stride, pad, act, i, cp, op, has_bias, wshape, zbg, act_param = ctx.cfg
    REC.append(dict(dy=ops.to_nchw(dy.detach()).cpu().double(), x=ops.to_nchw(xr.detach()).cpu().double()[:, :i],
                    w=w.detach().permute(0, 3, 1, 2)[:, :i].cpu().double(), y=None if y is None else ops.to_nchw(y.detach()).cpu().double(),
                    dx=None if out[0] is None else ops.to_nchw(out[0].detach()).cpu().double(),
                    dw=None if out[1] is None else out[1].detach().cpu().double(), stride=stride, pad=pad, act=act,
                    act_param=act_param, wshape=wshape))
    return out


ops._Conv2dFn.backward = staticmethod(traced)
model = SamsModel(hp)
model.load_state_dict(sd, strict=True)
model = model.cuda().train()
for p in model.parameters():
    p.requires_grad_(False)
for p in model.generator.parameters():
    p.requires_grad_(True)
db = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
res = model.training_step(db, 0, 0)
res.logs["loss/G/l1"].sum().backward()
for n, r in enumerate(REC):
    dy = r["dy"]
    o = r["wshape"][0]
    dy = dy[:, :o]
    if r["act"] != 0:
        yy = r["y"][:, :o]
        dy = dy * torch.where(yy > 0, torch.ones_like(yy), torch.full_like(yy, r["act_param"] if r["act"] == 2 else 0.0))
    msg = f"{n:3d} w{r['wshape']} x{tuple(r['x'].shape)} s{r['stride']} p{r['pad']}"
    if r["dx"] is not None:
        ref = torch.nn.grad.conv2d_input(r["x"].shape, r["w"][:o], dy, stride=r["stride"], padding=r["pad"])
        msg += f"  dx err {(r['dx'][:, :ref.shape[1]] - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30):.1e}"
    if r["dw"] is not None:
        ref = torch.nn.grad.conv2d_weight(r["x"], r["w"][:o].shape, dy, stride=r["stride"], padding=r["pad"])
        msg += f"  dw err {(r['dw'][:o, :ref.shape[1]] - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30):.1e}"
    print(msg)

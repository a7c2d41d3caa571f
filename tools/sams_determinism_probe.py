"""Diagnostic: which generator gradients differ between two identical SamsModel generator steps (full size, bs = 1)?"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from shineon_virtual_tryon_amd.data import synthetic_batch  # noqa: E402
from shineon_virtual_tryon_amd.sams_model import SamsModel  # noqa: E402

kw = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    kw[k] = eval(v)
hp = bench.sams_hparams(**kw)
torch.manual_seed(420)
model = SamsModel(hp).cuda().train()
batch = synthetic_batch(1, "cuda", seed=420, n_frames=hp.n_frames_total, height=hp.fine_height, width=hp.fine_width)
for p in model.parameters():
    p.requires_grad_(False)
for p in model.generator.parameters():
    p.requires_grad_(True)
buffers = copy.deepcopy({k: v for k, v in model.state_dict().items() if not k.startswith("criterion_VGG")})


def run(term):
    model.load_state_dict(buffers, strict=False)
    model.zero_grad(set_to_none=True)
    res = model.training_step(batch, 0, 0)
    res.logs[term].sum().backward()
    return {k: p.grad.clone() for k, p in model.generator.named_parameters() if p.grad is not None}


run("loss")
for term in ("loss/G/l1", "loss/G/vgg"):
    a, b = run(term), run(term)
    rows = sorted(((float((a[k] - b[k]).abs().max()) / max(float(a[k].abs().max()), 1e-30), k) for k in a), reverse=True)
    print(term, "differing tensors:", sum(r[0] > 0 for r in rows), "of", len(rows))
    for r in rows[:8]:
        print("   %.2e  %s" % r)

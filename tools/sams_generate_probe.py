"""Diagnostic: gradient of <w, frame_k> through generate_n_frames only (no discriminators / VGG), GPU vs fp64 oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sams_helpers as sh  # noqa: E402
from oracle import sams_oracle as so  # noqa: E402
from oracle.procedural import procedural_state_dict  # noqa: E402

import shineon_virtual_tryon_amd  # noqa: E402,F401
from shineon_virtual_tryon_amd import ops  # noqa: E402
from shineon_virtual_tryon_amd.data import synthetic_batch  # noqa: E402
from shineon_virtual_tryon_amd.sams_model import SamsModel  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "progressive"
g = sh.load_golden(tag)
hp = sh.sams_hparams(**sh.SAMS_VARIANTS[tag])
sd = procedural_state_dict(sh.golden_shapes(g))
batch = synthetic_batch(2, "cpu", height=hp.fine_height, width=hp.fine_width, n_frames=hp.n_frames_total, smooth=True)
torch.manual_seed(3)
wts = torch.randn(2, hp.n_frames_total, 3, hp.fine_height, hp.fine_width)
for which in ("last", "all"):
    osd = {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for k in so.optimizer_groups(osd)["generator"]:
        osd[k].requires_grad_(True)
    ob = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    _, _, frames = so.SamsOracle(osd, hp).generate_n_frames(ob)
    sel = slice(-1, None) if which == "last" else slice(None)
    (frames[:, sel] * wts[:, sel].double()).sum().backward()
    model = SamsModel(hp)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    for p in model.parameters():
        p.requires_grad_(False)
    for p in model.generator.parameters():
        p.requires_grad_(True)
    db = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    _, _, fr = model.generate_n_frames(db)
    loss = 0
    for f, t in enumerate(fr):
        if t is None or (which == "last" and f != len(fr) - 1):
            continue
        loss = loss + ops.tensor_sum(ops.blend(ops.fill_(torch.empty_like(ops.to_rows(t)), 0.0), t, None) if False else t * ops.to_rows(wts[:, f].cuda()))
    loss.backward()
    rows = []
    for k, p in model.generator.named_parameters():
        ref = osd["generator." + k].grad
        if p.grad is None or ref is None or ref.abs().max() < 1e-9:
            continue
        rows.append(((p.grad.double().cpu() - ref).abs().max().item() / ref.abs().max().item(), k))
    rows.sort(reverse=True)
    print(which, rows[:4])

"""Diagnostic: trace every custom backward (inputs / outputs checksums, strides) for the L1 term, in two scenarios that must
agree: A = generate + L1 only; B = the whole generator_step forward, then backward of the L1 term only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sams_helpers as sh  # noqa: E402
from oracle.procedural import procedural_state_dict  # noqa: E402

import shineon_virtual_tryon_amd  # noqa: E402,F401
from shineon_virtual_tryon_amd import ops, ops_sams  # noqa: E402
from shineon_virtual_tryon_amd.data import synthetic_batch  # noqa: E402
from shineon_virtual_tryon_amd.sams_model import SamsModel  # noqa: E402

tag = "progressive"
g = sh.load_golden(tag)
hp = sh.sams_hparams(**sh.SAMS_VARIANTS[tag])
sd = procedural_state_dict(sh.golden_shapes(g))
batch = synthetic_batch(2, "cpu", height=hp.fine_height, width=hp.fine_width, n_frames=hp.n_frames_total, smooth=True)
LOG = []


def desc(t):
    if not torch.is_tensor(t):
        return None
    d = t.detach().double()
    return (tuple(t.shape), tuple(t.stride()), round(d.sum().item(), 6), round(d.abs().sum().item(), 6))


def wrap(cls):
    orig = cls.backward

    def traced(ctx, *grads):
        out = orig(ctx, *grads)
        outs = out if isinstance(out, tuple) else (out,)
        LOG.append((cls.__name__, [desc(g_) for g_ in grads], [desc(o) for o in outs]))
        return out

    cls.backward = staticmethod(traced)


for mod in (ops, ops_sams):
    for name in dir(mod):
        obj = getattr(mod, name)
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
            wrap(obj)


def run(scenario):
    LOG.clear()
    model = SamsModel(hp)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    for p in model.parameters():
        p.requires_grad_(False)
    for p in model.generator.parameters():
        p.requires_grad_(True)
    db = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    if scenario == "A":
        fake, _, frames = model.generate_n_frames(db)
        loss = ops.l1_loss(frames[-1], ops.to_rows(db["image"][:, -1])) * model.wt_l1
    else:
        res = model.training_step(db, 0, 0)
        loss = res.logs["loss/G/l1"]
    loss.sum().backward()
    torch.cuda.synchronize()
    return list(LOG)


a, b = run("A"), run("B")
print("calls", len(a), len(b))
for i, (x, y) in enumerate(zip(a, b)):
    if x != y:
        print("first divergence at call", i, x[0], y[0])
        print(" A in ", x[1]); print(" B in ", y[1]); print(" A out", x[2]); print(" B out", y[2])
        for j in range(max(0, i - 3), i):
            print("  before:", j, a[j][0], a[j][2])
        break
else:
    print("identical")

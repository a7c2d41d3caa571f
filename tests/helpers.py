"""Shared helpers for the parity tests (may import oracle/: tests are the checker)."""
import os

import numpy as np
import torch

from oracle import shineon_oracle as oracle
from oracle.procedural import procedural_state_dict, shapes_of

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def checksums(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], dtype=np.float64)


def strided(t, s=8):
    return t.detach().cpu()[..., ::s, ::s].contiguous().numpy()


def assert_close(a, b, atol, rtol=0.0, what=""):
    a = torch.as_tensor(np.asarray(a)) if not isinstance(a, torch.Tensor) else a.detach().cpu()
    b = torch.as_tensor(np.asarray(b)) if not isinstance(b, torch.Tensor) else b.detach().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    diff = (a.double() - b.double()).abs()
    tol = atol + rtol * b.double().abs()
    bad = diff > tol
    assert not bad.any(), (f"{what}: max abs diff {diff.max().item():.3e} (atol {atol}, rtol {rtol}), "
                           f"{int(bad.sum())}/{bad.numel()} out of tolerance")


def assert_checksums(t, ref_cs, rel, what="", floor=1e-5):
    cs = checksums(t)
    # sum may cancel: compare it against the abs-sum scale
    # (gradients that are analytically zero, e.g. a conv bias in front of BatchNorm, are pure round-off:
    #  `floor` keeps them from being compared relatively)
    scale = max(abs(ref_cs[1]), 1e-30)
    assert abs(cs[0] - ref_cs[0]) <= rel * scale + floor, f"{what}: sum {cs[0]} vs {ref_cs[0]}"
    assert abs(cs[1] - ref_cs[1]) <= rel * scale + floor, f"{what}: abs-sum {cs[1]} vs {ref_cs[1]}"
    l2 = max(abs(ref_cs[2]), 0.0) ** 0.5
    assert abs(cs[2] - ref_cs[2]) <= 2 * rel * l2 * l2 + 2 * floor * l2 + floor * floor, f"{what}: sq-sum {cs[2]} vs {ref_cs[2]}"


def assert_grad_samples(get_grad, g, prefix, rel, what="", floor=2e-7, report_above=None):
    """Element-wise pin of gradients against the reference's own: the golden holds every N-th element (flattened OIHW order,
    key `<prefix><name>`, N encoded in the prefix as gs<N>:) of every parameter gradient of the reference's backward pass.
    |ours - ref| <= rel * max|ref| + floor per tensor (floor: gradients that are analytically zero - a bias in front of a
    normalisation - hold round-off noise of ~1e-9 in the reference).  Returns the number of tensors compared.
    report_above: print every tensor's measured error relative to its largest reference entry and how many of them exceed
    this value (`pytest -s`), so that a loose `rel` is a measured number per tensor and not a blanket."""
    import re

    stride = int(re.match(r"gs(\d+)", prefix).group(1)) if re.match(r"gs(\d+)", prefix) else 97
    keys = [k for k in (g.files if hasattr(g, "files") else g) if k.startswith(prefix)]
    bad, measured = [], []
    for k in keys:
        ref = np.asarray(g[k], dtype=np.float64)
        got = get_grad(k[len(prefix):]).detach().cpu().contiguous().reshape(-1)[::stride].double().numpy()
        assert got.shape == ref.shape, (what, k, got.shape, ref.shape)
        err, big = np.abs(got - ref).max(), np.abs(ref).max()
        measured.append((max(err - floor, 0.0) / big if big > 0 else 0.0, k[len(prefix):]))
        if err > rel * big + floor:
            bad.append(f"{k[len(prefix):]}: {err:.3e} vs max {big:.3e}")
    if report_above is not None:
        over = [m for m in measured if m[0] > report_above]
        print(f"[{what}] {len(over)} of {len(measured)} gradient tensors are further than {report_above:g} of their largest "
              f"entry from the reference (asserted bound {rel:g}):")
        for e, name in sorted(measured, reverse=True):
            print(f"    {e:9.3e}  {name}")
    assert not bad, f"{what}: {len(bad)}/{len(keys)} gradient tensors differ element-wise from the reference: " + "; ".join(bad[:8])
    if report_above is not None:
        return len(keys), sorted(measured, reverse=True)
    return len(keys)


def synthetic_cpu_batch(bs=2):
    from shineon_virtual_tryon_amd.data import synthetic_batch

    return synthetic_batch(bs, "cpu", smooth=True)


WARP_HP = dict(n_frames_total=1, person_inputs=["agnostic", "cocopose"], cloth_inputs=["cloth"], grid_size=5,
               fine_height=256, fine_width=192, ngf=64)


def unet_hp(**kw):
    hp = dict(n_frames_total=1, person_inputs=["agnostic", "densepose"], cloth_inputs=["cloth"], self_attn=False,
              num_attn=2, activation=None, flow_warp=False, pen_flow_mask=1.0)
    hp.update(kw)
    return hp


UNET_VARIANTS = {"plain": {}, "gelu": {"activation": "gelu"}, "attn": {"self_attn": True},
                 "attn_gelu": {"self_attn": True, "activation": "gelu"}}


def golden_state(npz, seed=420):
    """Procedural state_dict for the key/shape list recorded next to a golden."""
    keys = [str(k) for k in npz["state_keys"]]
    shapes = [eval(str(s)) for s in npz["state_shapes"]]  # noqa: S307 - our own fixture
    return procedural_state_dict(dict(zip(keys, shapes)), seed)


def make_namespace(is_train=True, **kw):
    import argparse

    base = dict(n_frames_total=1, person_inputs=["agnostic", "densepose"], cloth_inputs=["cloth"], is_train=is_train,
                ngf=64, grid_size=5, fine_height=256, fine_width=192, self_attn=False, num_attn=2, flow_warp=False,
                activation=None, display_count=10 ** 9, pen_flow_mask=1.0, lr=1e-4, keep_epochs=5, decay_epochs=5,
                batch_size=2, workers=0, dataset="synthetic", no_shuffle=True)
    base.update(kw)
    return argparse.Namespace(**base)

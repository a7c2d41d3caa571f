"""Regenerates the committed igemm plans file (shineon-virtual-tryon_amd/plans/gfx950.txt) on an MI355X.

    SHINEON_AUTOTUNE=2 SHINEON_PLANS=none python tools/make_plans.py [out.txt]

Every layer shape of the configurations the tests and bench.py run (BASELINE configs at bs = 1 / 2 / 4 / 8, the
n_frames = 3 and n_frames = 5 U-Nets, the SAMS-GAN's default networks at bs = 4 / 1) is seen once eagerly; in thorough mode each (tile, waves, split-K) candidate is
timed on a warmed-up chip, six launches each, and the fastest is kept.  The file makes the kernel choice - hence the
split-K summation order, hence the bit pattern of every result - the same in every process that loads it.
"""
import os
import sys

os.environ.setdefault("SHINEON_AUTOTUNE", "2")
os.environ.setdefault("SHINEON_PLANS", "none")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import shineon_virtual_tryon_amd as pkg  # noqa: E402
from shineon_virtual_tryon_amd.data import synthetic_batch  # noqa: E402
from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel  # noqa: E402
from shineon_virtual_tryon_amd.warp_model import WarpModel  # noqa: E402


def one_step(model, batch):
    (opt,), _ = model.configure_optimizers()
    opt.zero_grad()
    res = model.training_step(batch, 0)
    res.minimize.backward()
    opt.step()
    torch.cuda.synchronize()
    return model


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "shineon-virtual-tryon_amd", "plans", "gfx950.txt")
    dev = torch.device("cuda", 0)
    L = pkg.lib()
    jobs = []
    for bs in (4, 8, 2, 1):
        jobs.append(("warp", dict(person_inputs=["agnostic", "cocopose"]), dict(bs=bs)))
        for kw in (dict(self_attn=True, activation="gelu"), dict(self_attn=True, activation=None),
                   dict(self_attn=False, activation="gelu"), dict(self_attn=False, activation=None)):
            if bs in (4, 8) and kw != dict(self_attn=True, activation="gelu"):
                continue
            jobs.append(("unet", dict(person_inputs=["agnostic", "densepose"], **kw), dict(bs=bs)))
    jobs.append(("unet", dict(person_inputs=["agnostic", "densepose"], n_frames_total=5, flow_warp=True), dict(bs=1, n_frames=5)))
    jobs.append(("unet", dict(person_inputs=["agnostic", "densepose"], n_frames_total=5, flow_warp=True), dict(bs=2, n_frames=5)))
    jobs.append(("unet", dict(person_inputs=["agnostic", "densepose"], n_frames_total=3, flow_warp=True, self_attn=False,
                              fine_height=128, fine_width=64), dict(bs=2, n_frames=3, height=128, width=64)))
    for kind, hp, b in jobs:
        torch.manual_seed(0)
        cls = WarpModel if kind == "warp" else UnetMaskModel
        model = cls(bench.hparams(**hp)).to(dev).train()
        model.global_step = 1
        batch = synthetic_batch(b["bs"], dev, height=b.get("height", 256), width=b.get("width", 192),
                                n_frames=b.get("n_frames", 1), smooth=True)
        before = L.so_igemm_plan_count()
        one_step(model, batch)
        print(f"{kind} {hp} {b}: +{L.so_igemm_plan_count() - before} plans", flush=True)
        del model
        torch.cuda.empty_cache()
    # SAMS-GAN (bench.py --config sams): the reference-default networks at bs = 4 and bs = 1, one three-optimizer step each
    from shineon_virtual_tryon_amd.sams_model import SamsModel
    from shineon_virtual_tryon_amd.trainer import MultiOptimizerStep

    for bs in (4, 1):
        torch.manual_seed(0)
        hp = bench.sams_hparams()
        model = SamsModel(hp).to(dev).train()
        model.global_step = 1
        batch = synthetic_batch(bs, dev, n_frames=hp.n_frames_total)
        before = L.so_igemm_plan_count()
        MultiOptimizerStep(model, model.configure_optimizers()[0])(batch, 0)
        torch.cuda.synchronize()
        print(f"sams bs={bs}: +{L.so_igemm_plan_count() - before} plans", flush=True)
        del model
        torch.cuda.empty_cache()
    n = L.so_igemm_plans_save(out.encode())
    print(f"saved {n} plans to {out}")


if __name__ == "__main__":
    main()

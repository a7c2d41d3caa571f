"""Trainer.fit / Trainer.fit_chained throughput on the synthetic dataset (HBM-resident, graph + overlap) next to what
bench.py reports for the same step: VERDICT r1 item 3 (the fast schedule behind the product API)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from shineon_virtual_tryon_amd.options import TrainOptions  # noqa: E402
from shineon_virtual_tryon_amd.registry import find_model_using_name  # noqa: E402
from shineon_virtual_tryon_amd.trainer import DeviceBatches, Trainer  # noqa: E402
from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel  # noqa: E402
from shineon_virtual_tryon_amd.warp_model import WarpModel  # noqa: E402

dev = torch.device("cuda", 0)
for name, extra in (("gmm", []), ("tom", ["--self_attn", "--activation", "gelu", "--allow_random_vgg"])):
    opt = TrainOptions().parse(["--model", name, "--dataset", "synthetic", "--name", "t", "-b", "4", "--workers", "0",
                                "--synthetic_length", "128", "--experiments_dir", "/tmp/trainer_speed", "--no_shuffle"] + extra,
                               interactive=False)
    model = find_model_using_name(opt.model)(opt)
    tr = Trainer(default_root_dir="/tmp/trainer_speed", max_epochs=3, limit_val_batches=1, val_check_interval=10 ** 6)
    tr.fit(model)            # epoch 0 includes dataset upload, plan lookup, graph capture
    ep = tr.epoch_seconds
    print(f"Trainer.fit --model {name}: epochs of 32 steps (bs 4) took {[round(e, 3) for e in ep]} s -> steady "
          f"{1e3 * min(ep[1:]) / 32:.2f} ms/step ({4 * 32 / min(ep[1:]):.1f} frames/s)", flush=True)

# chained: fit_chained over HBM-resident batches
warp = WarpModel(bench.hparams(person_inputs=["agnostic", "cocopose"]))
unet = UnetMaskModel(bench.hparams(person_inputs=["agnostic", "densepose"]))
warp.global_step = unet.global_step = 1
opt = TrainOptions().parse(["--model", "gmm", "--dataset", "synthetic", "--name", "t", "-b", "4", "--workers", "0",
                            "--synthetic_length", "64"], interactive=False)
from shineon_virtual_tryon_amd.data import SyntheticDataset  # noqa: E402

batches = DeviceBatches(SyntheticDataset(opt), 4, dev, shuffle=False, keys=warp.batch_keys() | unet.batch_keys())
tr = Trainer(graph=True, overlap=True)
eng = tr.fit_chained(warp, unet, batches, steps=16)   # builds the engine (capture) + 16 steps
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
for _ in range(5):
    for b in batches:
        eng(b)
        n += 1
eng.synchronize()
dt = time.perf_counter() - t0
print(f"Trainer.fit_chained engine over HBM-resident batches: {1e3 * dt / n:.2f} ms/step ({4 * n / dt:.1f} frames/s) incl. batch gather + copy")

"""How far theta / the TPS grid / the warped cloth of the HIP WarpModel sit from (a) the REFERENCE's own fp32 CPU values
(tests/golden/warp_model.npz, bs=2) and (b) the oracle's fp32 and fp64 values at bs=4 (tests/golden/full/chain_bs4.npz)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import shineon_virtual_tryon_amd  # noqa: E402,F401
import fullsize_cases as fc  # noqa: E402
import gradfix as gf  # noqa: E402
from helpers import golden_state, load_golden, make_namespace, strided, synthetic_cpu_batch  # noqa: E402
from shineon_virtual_tryon_amd.warp_model import WarpModel  # noqa: E402

dev = torch.device("cuda", 0)
g = load_golden("warp_model.npz")
hp = make_namespace(person_inputs=["agnostic", "cocopose"])
model = WarpModel(hp)
model.load_state_dict(golden_state(g), strict=True)
model = model.to(dev).train()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_cpu_batch(2).items()}
person = torch.cat([batch[k] for k in hp.person_inputs], 1)
with torch.no_grad():
    grid, theta = model(person, batch["cloth"])
    from shineon_virtual_tryon_amd import ops
    warped = ops.grid_sample(batch["cloth"], grid, "border")
print("bs=2 vs REFERENCE golden: theta %.3e  grid(s8) %.3e  warped(s8) %.3e" % (
    np.abs(theta.cpu().numpy() - g["theta"]).max(), np.abs(strided(grid.permute(0, 3, 1, 2)) - g["grid_s8"]).max(),
    np.abs(strided(warped) - g["warped_cloth_s8"]).max()))

fix = gf.load("chain_bs4")
m4, _ = fc.build_warp(dev)
b4 = fc.to_device(fc.smooth_batch(4), dev)
p4 = torch.cat([b4[k] for k in fc.WHP["person_inputs"]], 1)
with torch.no_grad():
    grid, theta = m4(p4, b4["cloth"])
    warped = ops.grid_sample(b4["cloth"], grid, "border")
t32, t64 = fix["warp:theta32"], fix["warp:theta64"]
th = theta.cpu().double().numpy()
print("bs=4 theta: ours-fp32 %.3e  ours-fp64 %.3e  fp32-fp64 %.3e" % (np.abs(th - t32).max(), np.abs(th - t64).max(), np.abs(t32 - t64).max()))
gf.check_output(fix, "warp:grid", grid.permute(0, 3, 1, 2), 1.0, "probe", either=False)
gf.check_output(fix, "warp:warped_cloth", warped, 1.0, "probe", either=False)
# TPS alone on the ORACLE's fp32 theta: separates the TPS evaluation from the upstream theta difference
from oracle import shineon_oracle as oracle  # noqa: E402  (probe = checker side)
c = oracle.tps_constants(256, 192, 5)
consts = tuple(c[k].contiguous().to(dev) for k in ("Li", "px", "py", "gx", "gy"))
with torch.no_grad():
    g_ours = ops.tps_grid(torch.from_numpy(t32).float().to(dev), consts, 256, 192, 25)
gf.check_output(fix, "warp:grid", g_ours.permute(0, 3, 1, 2), 1.0, "probe: OUR tps on the ORACLE's fp32 theta", either=False)

"""Child process of tests/test_00_multi_rank_gpu.py: one rank of a 2-rank data-parallel run (started as a fresh python process,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment; backend nccl = RCCL on two GPUs, or gloo with both ranks on
one GPU).

Checks (reference behaviour: Lightning DDP configured at train.py:76-85, DistributedSampler at base_model.py:113-121):
  1. 2 ranks x bs=2 give, after the mean all-reduce, the gradients of 1 rank x bs=4 (InstanceNorm and attention are
     per-sample, every loss is a mean over the batch) - element-wise, up to fp32 reduction order;
  2. after 3 optimisation steps through trainer.TrainStep (hipGraph + asynchronous exchange) the parameters of the
     two ranks are bit-identical;
  3. BatchNorm running statistics stay per rank without the buffer broadcast and equal rank 0's with it;
  4. SAMS-GAN: two iterations of trainer.MultiOptimizerStep (three optimizers, one gradient exchange each) leave the two
     ranks bit-identical, with the step counters and the batch-mean losses of a single-rank full-batch run.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import shineon_virtual_tryon_amd  # noqa: E402,F401
from helpers import make_namespace  # noqa: E402
from oracle.procedural import procedural_state_dict, shapes_of  # noqa: E402
from shineon_virtual_tryon_amd.data import synthetic_batch  # noqa: E402
from shineon_virtual_tryon_amd.trainer import (GradientAllReducer, TrainStep, broadcast_parameters,  # noqa: E402
                                               init_distributed)
from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel  # noqa: E402
from shineon_virtual_tryon_amd.warp_model import WarpModel  # noqa: E402

rank, world = init_distributed()  # nccl (= RCCL) on two GPUs; gloo when both ranks share one GPU (SHINEON_DIST_BACKEND)
assert world == 2
dev = torch.device("cuda", torch.cuda.current_device())


def build(cls, **hp):
    m = cls(make_namespace(allow_random_vgg=True, lr=1e-3, **hp))
    m.load_state_dict(procedural_state_dict(shapes_of(m.state_dict())))
    m.global_step = 1
    return m.to(dev).train()


def take(batch, lo, hi):
    return {k: (v[lo:hi].contiguous() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


full = synthetic_batch(4, dev, smooth=True)
mine = take(full, 2 * rank, 2 * rank + 2)

# ---- 1. averaged half-batch gradients == full-batch gradients -------------------------------------------------------
unet = build(UnetMaskModel, self_attn=True, activation="gelu")
(opt,), _ = unet.configure_optimizers()
opt.zero_grad()
unet.training_step(mine, 0).minimize.backward()
scale = GradientAllReducer(opt.flat_grads).all_reduce()
torch.cuda.synchronize()
avg = opt.flat_grads.clone() * scale
solo = build(UnetMaskModel, self_attn=True, activation="gelu")
(opt1,), _ = solo.configure_optimizers()
opt1.zero_grad()
solo.training_step(full, 0).minimize.backward()
torch.cuda.synchronize()
for (name, p), (_, q) in zip(unet.named_parameters(), solo.named_parameters()):
    if not p.requires_grad:
        continue
    g2, g1 = p.grad * scale, q.grad
    tol = 2e-3 * float(g1.abs().max()) + 2e-7   # different batch => different tiles / split-K order: the parity tolerance
    if g1.numel() == 1:   # attention gamma: one heavily cancelling dot product (kappa ~ 1e3), see test_parity_bs4_gpu.py
        tol = max(tol, 0.2 * float(g1.abs().max()) + 1e-6)
    err = float((g2 - g1).abs().max())
    assert err <= tol, f"rank {rank}: {name}: 2x2 vs 1x4 gradient differs by {err} (tol {tol})"
print(f"DP_GRAD_OK {rank}", flush=True)

# ---- 2. three steps through the product's step engine: ranks stay bit-identical ----------------------------------------
torch.manual_seed(100 + rank)          # different initial weights per rank: the broadcast must fix that
unet2 = UnetMaskModel(make_namespace(allow_random_vgg=True, lr=1e-3, self_attn=True, activation="gelu")).to(dev).train()
unet2.global_step = 1
(opt2,), _ = unet2.configure_optimizers()
broadcast_parameters(unet2, optimizer=opt2)
for p in unet2.criterionVGG.parameters():  # frozen, not in the optimizer slab: same everywhere via its own broadcast
    dist.broadcast(p.data, 0)
engine = TrainStep(unet2, opt2, mine, graph=True, overlap=True)
for i in range(3):
    engine(take(synthetic_batch(4, dev, smooth=True, start=4 * i), 2 * rank, 2 * rank + 2))
engine.flush()
torch.cuda.synchronize()
both = [torch.empty_like(opt2.flat_params) for _ in range(2)]
dist.all_gather(both, opt2.flat_params)
assert torch.equal(both[0], both[1]), f"parameters diverged: {float((both[0] - both[1]).abs().max())}"
assert opt2._steps == 3
print(f"DP_STEP_OK {rank}", flush=True)

# ---- 3. BatchNorm statistics: per rank without the broadcast, rank 0's with it -------------------------------------------
for sync in (False, True):
    warp = build(WarpModel, person_inputs=["agnostic", "cocopose"])
    (optw,), _ = warp.configure_optimizers()
    broadcast_parameters(warp, optimizer=optw)
    eng = TrainStep(warp, optw, mine, graph=True, overlap=True, sync_buffers=sync)
    for i in range(2):
        eng(mine)
    eng.flush()
    if sync:   # the broadcast happens at the START of a step (DDP: before the forward); make the last update visible
        from shineon_virtual_tryon_amd.trainer import broadcast_buffers

        broadcast_buffers(warp)
    torch.cuda.synchronize()
    rm = warp.extractionA.model[2].running_mean
    got = [torch.empty_like(rm) for _ in range(2)]
    dist.all_gather(got, rm.contiguous())
    same = torch.equal(got[0], got[1])
    assert same == sync, f"sync_buffers={sync}: running_mean equal across ranks = {same}"
    gp = [torch.empty_like(optw.flat_params) for _ in range(2)]
    dist.all_gather(gp, optw.flat_params)
    assert torch.equal(gp[0], gp[1])
print(f"DP_BN_OK {rank}", flush=True)

# ---- 4. SAMS-GAN: the three optimizers of trainer.MultiOptimizerStep, each with its own gradient exchange -----------------
# Per-sample normalisations everywhere (SPADE over InstanceNorm, discriminators spectral + InstanceNorm) make 2 ranks x bs 1
# equivalent to 1 rank x bs 2; the batch-norm default would not be (per-rank statistics, as in the reference under DDP).
import sams_helpers as sh  # noqa: E402
from shineon_virtual_tryon_amd.sams_model import SamsModel  # noqa: E402
from shineon_virtual_tryon_amd.trainer import MultiOptimizerStep  # noqa: E402

shp = sh.sams_hparams(norm_G="spectralspadeinstance3x3", norm_D="spectralinstance", activation="gelu", allow_random_vgg=True)
sfull = synthetic_batch(2, dev, height=shp.fine_height, width=shp.fine_width, n_frames=shp.n_frames_total, smooth=True)
smine = take(sfull, rank, rank + 1)


def sams():
    m = SamsModel(shp)
    m.load_state_dict(procedural_state_dict(shapes_of(m.state_dict())))
    m.global_step = 1
    return m.to(dev).train()


pair, solo = sams(), sams()
popts, sopts = pair.configure_optimizers()[0], solo.configure_optimizers()[0]
pstep, sstep = MultiOptimizerStep(pair, popts), MultiOptimizerStep(solo, sopts)
assert pstep.reducers is not None and len(pstep.reducers) == 3
sstep.reducers = None                      # the single-rank reference run exchanges nothing
for it in range(2):
    pres, sres = pstep(smine, it), sstep(sfull, it)
torch.cuda.synchronize()
for name, po, so_ in zip(("generator", "multiscale D", "temporal D"), popts, sopts):
    a, b = po.flat_params, so_.flat_params
    both = [torch.empty_like(a) for _ in range(2)]
    dist.all_gather(both, a)
    assert torch.equal(both[0], both[1]), f"SAMS {name}: ranks diverged"
    # two Adam steps from identical weights: the 2-rank run and the 1-rank full-batch run took the same direction
    assert po._steps == so_._steps == 2
for k in pres[0].logs:   # the per-rank generator loss differs (different sample), the mean of the two is the full-batch one
    mine_v = pres[0].logs[k].detach().reshape(1).float()
    got = [torch.empty_like(mine_v) for _ in range(2)]
    dist.all_gather(got, mine_v)
    mean = 0.5 * (got[0] + got[1])
    ref = sres[0].logs[k].detach().reshape(1).float()
    if k in ("loss/G/l1", "loss/G/vgg", "loss/G/l1+vgg"):
        assert abs(float(mean - ref)) <= 2e-3 * max(1.0, abs(float(ref))), (k, float(mean), float(ref))
print(f"DP_SAMS_OK {rank}", flush=True)
dist.barrier()
dist.destroy_process_group()
print(f"DP_ALL_OK {rank}", flush=True)

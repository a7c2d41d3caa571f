"""Child process of tests/test_00_multi_rank_gpu.py: the gradient-exchange path over RCCL on a one-GPU box.

RCCL refuses two ranks on one device, so the 2-rank tests on the 1-GPU box use gloo.  This child instead opens a ONE-rank nccl
(= RCCL) group (SHINEON_SINGLE_RANK_GROUP=1, trainer.init_distributed) and runs trainer.TrainStep with every collective issued
for real: parameter / buffer broadcast, the per-bucket all-reduce on the communication stream behind the signal nodes of the
captured backward pass (BucketedExchange), Adam per bucket.  With one rank the sums are the inputs, so three steps must leave
parameters and Adam moments BIT-IDENTICAL to the same steps without a process group."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import shineon_virtual_tryon_amd  # noqa: E402,F401
from helpers import make_namespace  # noqa: E402
from shineon_virtual_tryon_amd import trainer  # noqa: E402
from shineon_virtual_tryon_amd.data import synthetic_batch  # noqa: E402
from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel  # noqa: E402
from shineon_virtual_tryon_amd.warp_model import WarpModel  # noqa: E402

dev = torch.device("cuda", 0)


def run(which, graph, bucket_bytes=16 << 20):
    torch.manual_seed(5)
    if which == "unet":
        model = UnetMaskModel(make_namespace(self_attn=True, activation="gelu", allow_random_vgg=True, lr=1e-3))
    else:
        model = WarpModel(make_namespace(person_inputs=["agnostic", "cocopose"], lr=1e-3))
    model = model.to(dev).train()
    model.global_step = 1
    (opt,), _ = model.configure_optimizers()
    trainer.broadcast_parameters(model, optimizer=opt)
    batches = [synthetic_batch(2, dev, smooth=True, start=2 * i) for i in range(3)]
    eng = trainer.TrainStep(model, opt, batches[0], graph=graph, overlap=True,
                            **({} if bucket_bytes is None else {"bucket_bytes": bucket_bytes}))
    for b in batches:
        eng(b)
    eng.flush()
    torch.cuda.synchronize()
    if which == "warp":
        assert model.tower_streams, "the stand-alone WarpModel runs its two extraction towers on two streams by default"
    return opt, eng


def run_chained(bucketed):
    """trainer.ChainedTrainStep, two-stream schedule: bucketed exchange per model (default with collectives) or whole-slab."""
    torch.manual_seed(6)
    warp = WarpModel(make_namespace(person_inputs=["agnostic", "cocopose"], lr=1e-3)).to(dev).train()
    unet = UnetMaskModel(make_namespace(self_attn=True, activation="gelu", allow_random_vgg=True, lr=1e-3)).to(dev).train()
    warp.global_step = unet.global_step = 1
    (optw,), _ = warp.configure_optimizers()
    (optu,), _ = unet.configure_optimizers()
    batches = [synthetic_batch(2, dev, smooth=True, start=2 * i) for i in range(3)]
    eng = trainer.ChainedTrainStep(warp, optw, unet, optu, batches[0], schedule="pipeline", bucketed=bucketed, bucket_bytes=16 << 20)
    for b in batches:
        eng(b)
    eng.synchronize()
    return (optw, optu), eng


os.environ["SHINEON_SINGLE_RANK_GROUP"] = "0"
chained_plain, _ = run_chained(False)
plain = {(w, g): run(w, g) for w in ("unet", "warp") for g in (True, False)}
# the DEFAULT bucket size (64 MB): the warp model's 76 MB slab falls into two buckets, and the first one holds all of
# extractionA plus the first layers of extractionB - gradients produced on BOTH tower streams (BucketedExchange._signal)
plain.update({("warp", g, None): run("warp", g, None) for g in (True, False)})
assert all(e.exchange is None for _, e in plain.values()) and not trainer._collective()

os.environ["SHINEON_SINGLE_RANK_GROUP"] = "1"
rank, world = trainer.init_distributed()
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl" and trainer._collective()
print("RCCL", ".".join(str(v) for v in torch.cuda.nccl.version()), flush=True)
probe = torch.arange(8, dtype=torch.float32, device=dev)
dist.all_reduce(probe)
torch.cuda.synchronize()
assert torch.equal(probe.cpu(), torch.arange(8, dtype=torch.float32))
for key, (o0, _) in plain.items():
    o1, e1 = run(*key)
    assert e1.exchange is not None and e1.exchange.active and len(e1.exchange.buckets) >= 2, key
    print(key, e1.exchange.describe(), flush=True)
    if len(key) == 3:   # default bucket size: at least one bucket's gradients come from more than one stream
        multi = [b for b, st in e1.exchange._streams.items() if len(st) > 1]
        assert multi, ("no bucket straddles the two tower streams - the test no longer covers the multi-stream signal", key)
        print(key, "buckets fed by several streams:", multi, flush=True)
    assert o1._steps == o0._steps == 3
    for a, b in zip(o1._flat, o0._flat):
        if a is o1._flat[1]:
            continue   # the gradient slab: scratch
        assert torch.equal(a, b), (key, float((a - b).abs().max()))
for bucketed in (False, True):
    opts, eng = run_chained(bucketed)
    assert (eng.exu is not None) == bucketed and eng.redu.active
    for o1, o0 in zip(opts, chained_plain):
        assert o1._steps == o0._steps == 3
        for a, b in zip(o1._flat, o0._flat):
            if a is o1._flat[1]:
                continue
            assert torch.equal(a, b), ("chained", bucketed, float((a - b).abs().max()))
    print("chained", "bucketed" if bucketed else "whole-slab", "one communicator", "bit-identical", flush=True)
dist.barrier()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK", flush=True)

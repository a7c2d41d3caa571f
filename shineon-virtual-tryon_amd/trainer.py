"""Lightning-free training / test driver exposing the hooks the reference's Trainer calls
(reference: train.py:32-141; Lightning 0.9 `Trainer(gpus, distributed_backend="ddp", ...)`).

One process per GPU.  Data parallelism = mean all-reduce of the flat gradient slab over RCCL
(torch.distributed backend "nccl"), issued in a few large buckets on a side stream so that the
transfer over xGMI overlaps the remaining work of the step; BatchNorm statistics stay per-rank like the
reference's nn.BatchNorm2d under DDP, with rank 0's running stats broadcast to mirror DDP's
broadcast_buffers.
"""
import logging
import os
import os.path as osp
import signal

import torch
import torch.distributed as dist

from .options import str2num

logger = logging.getLogger("logger")


def init_distributed(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* if launched under torchrun."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return dist.get_rank() if dist.is_initialized() else 0, max(world, 1)
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


class GradientAllReducer:
    """Mean all-reduce of a flat gradient slab in `n_buckets` contiguous pieces.

    xGMI is point-to-point (7 links per GPU), so a few large messages are preferred to DDP's many 25 MB
    buckets: the 76-91 MB slab is cut into 4 pieces by default."""

    def __init__(self, flat_grads, n_buckets=4, group=None):
        self.flat = flat_grads
        self.group = group
        n = flat_grads.numel()
        step = (n + n_buckets - 1) // n_buckets
        step = (step + 1023) // 1024 * 1024
        self.buckets = [flat_grads[i:min(i + step, n)] for i in range(0, n, step)]
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

        self._works = None

    def start(self):
        """Issue the asynchronous all-reduce of every bucket (RCCL runs on its own stream, ordered after the work
        already queued on the current stream), so that it overlaps whatever is launched next."""
        if self.world > 1:
            self._works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets]

    def finish(self):
        """Make the current stream wait for the reduction; returns the factor the optimizer must scale the
        (summed) gradients by: 1 / world."""
        if self._works is not None:
            for w in self._works:
                w.wait()
            self._works = None
        return 1.0 / self.world

    def all_reduce(self):
        """start() + finish(): returns the factor the optimizer must scale gradients by (1 / world)."""
        if self.world == 1:
            return 1.0
        self.start()
        return self.finish()


def broadcast_buffers(model, src=0):
    """DDP's broadcast_buffers=True: rank 0's BatchNorm running stats win."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for b in model.buffers():
        dist.broadcast(b, src)


def broadcast_parameters(model, src=0):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for p in model.parameters():
        dist.broadcast(p.data, src)


def _to_device(batch, device):
    return {k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


class Trainer:
    def __init__(self, gpus=None, distributed_backend="ddp", precision=32, default_root_dir="experiments",
                 accumulate_grad_batches=1, max_epochs=10, val_check_interval=1.0, limit_train_batches=1.0,
                 limit_val_batches=1.0, fast_dev_run=False, save_count=10000, resume_from_checkpoint=None,
                 broadcast_bn_buffers=True, **_):
        self.rank, self.world = init_distributed()
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if self.device is None:
            raise RuntimeError("the MI355X trainer needs a GPU; there is no CPU fallback (see oracle/ for checks)")
        self.root = default_root_dir
        self.accumulate = max(1, accumulate_grad_batches)
        self.max_epochs = 1 if fast_dev_run else max_epochs
        self.val_check_interval = val_check_interval
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.fast_dev_run = fast_dev_run
        self.save_count = save_count
        self.resume = resume_from_checkpoint
        self.broadcast_bn_buffers = broadcast_bn_buffers
        self.global_step = 0
        self.current_epoch = 0
        self.model = self.optimizer = self.scheduler = None

    # ---- checkpoints (Lightning-style dict so reference checkpoints' state_dict drops in) --------------
    def save_checkpoint(self, path):
        if self.rank != 0:
            return
        os.makedirs(osp.dirname(path) or ".", exist_ok=True)
        torch.save({
            "state_dict": {k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()},
            "hparams": vars(self.model.hparams),
            "optimizer_states": [self.optimizer.state_dict()] if self.optimizer else [],
            "global_step": self.global_step, "epoch": self.current_epoch,
        }, path)

    def _maybe_resume(self):
        if not self.resume:
            return
        ckpt = torch.load(self.resume, map_location="cpu", weights_only=False)
        self.global_step = ckpt.get("global_step", 0)
        self.current_epoch = ckpt.get("epoch", 0)
        if ckpt.get("optimizer_states") and self.optimizer is not None:
            self.optimizer.load_state_dict(ckpt["optimizer_states"][0])

    def _limit(self, n, lim):
        lim = str2num(str(lim))
        return min(n, lim) if isinstance(lim, int) else max(1, int(n * lim))

    # ---- fit ------------------------------------------------------------------------------------------
    def fit(self, model):
        self.model = model.to(self.device)
        model.trainer = self
        model.prepare_data()
        model.setup("fit")
        broadcast_parameters(model)
        (self.optimizer,), (self.scheduler,) = model.configure_optimizers()
        self._maybe_resume()
        reducer = GradientAllReducer(self.optimizer.flat_grads)
        train_loader, val_loader = model.train_dataloader(), model.val_dataloader()
        ckpt_dir = osp.join(self.root, "checkpoints")
        signal.signal(signal.SIGINT, lambda *a: (self.save_checkpoint(osp.join(ckpt_dir, "interrupted_by_Ctrl-C.ckpt")), exit()))
        n_train = self._limit(len(train_loader), self.limit_train_batches)
        vci = str2num(str(self.val_check_interval))
        val_every = vci if isinstance(vci, int) and vci > 0 else max(1, int(n_train * (vci or 1)))
        try:
            for epoch in range(self.current_epoch, self.max_epochs):
                self.current_epoch = model.current_epoch = epoch
                if hasattr(train_loader.sampler, "set_epoch"):
                    train_loader.sampler.set_epoch(epoch)
                model.train()
                self.optimizer.zero_grad()
                for i, batch in enumerate(train_loader):
                    if i >= n_train or (self.fast_dev_run and i >= 1):
                        break
                    if self.broadcast_bn_buffers:
                        broadcast_buffers(model)
                    model.global_step = self.global_step
                    result = model.training_step(_to_device(batch, self.device), i)
                    (result.minimize / self.accumulate).backward()
                    if (i + 1) % self.accumulate == 0:
                        scale = reducer.all_reduce()
                        self.optimizer.step(grad_scale=scale)
                        self.optimizer.zero_grad()
                        self.global_step += 1
                        if self.global_step % self.save_count == 0:
                            self.save_checkpoint(osp.join(ckpt_dir, f"step_{self.global_step:09d}.ckpt"))
                    if (i + 1) % val_every == 0:
                        self._validate(model, val_loader)
                        model.train()
                self.scheduler.step()
        except Exception as e:  # mirror train.py:63-66: checkpoint, then re-raise
            self.save_checkpoint(osp.join(ckpt_dir, f"interrupted_by_{type(e).__name__}.ckpt"))
            raise
        self.save_checkpoint(osp.join(ckpt_dir, "final.ckpt"))

    @torch.no_grad()
    def _validate(self, model, loader):
        model.eval()
        n = self._limit(len(loader), self.limit_val_batches)
        losses = []
        for i, batch in enumerate(loader):
            if i >= n or (self.fast_dev_run and i >= 1):
                break
            res = model.validation_step(_to_device(batch, self.device), i)
            losses.append(float(res.checkpoint_on))
        model.on_validation_epoch_end()
        return sum(losses) / max(1, len(losses))

    # ---- test -----------------------------------------------------------------------------------------
    @torch.no_grad()
    def test(self, model):
        self.model = model.to(self.device)
        model.trainer = self
        model.setup("test")
        model.eval()
        outs = []
        for i, batch in enumerate(model.test_dataloader()):
            outs.append(model.test_step(_to_device(batch, self.device), i))
        return outs

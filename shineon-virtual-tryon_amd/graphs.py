"""hipGraph capture of a model's forward + backward (`training_step` + `.backward()`).

One training step of the hot path is ~850 short kernel launches driven from Python; captured once into a
hipGraph (through torch.cuda.CUDAGraph, which records every launch made on the capture stream — including
the ctypes launches into libshineon_hip.so) it replays with a single host call.  The optimizer step and the
data-parallel all-reduce stay outside the graph (two launches, and RCCL is left uncaptured on purpose).

Constraints (same as any stream capture): static shapes, inputs copied into static buffers, no host
synchronisation inside `training_step`; gradients must already be views of the optimizer's flat slab.
"""
import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, sample_batch, warmup=2, alias_keys=()):
        """alias_keys: batch entries used in place (not cloned), e.g. a tensor that is itself the static output of
        another captured graph (the warped cloth handed from the warp stage to the try-on stage)."""
        self.model, self.optimizer = model, optimizer
        self.static_batch = {k: (v.clone() if isinstance(v, torch.Tensor) and k not in alias_keys else v)
                             for k, v in sample_batch.items()}
        self.alias_keys = tuple(alias_keys)
        self.result = None
        optimizer.zero_grad()  # plants the flat gradient views before anything is recorded
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._step()

    def _step(self):
        self.optimizer.zero_grad()
        res = self.model.training_step(self.static_batch, 0)
        res.minimize.backward()
        # keep only detached values: a live autograd graph would pin the parameters' AccumulateGrad nodes to the
        # stream of an earlier iteration and break the capture that follows
        res.minimize = res.minimize.detach()
        res.logs = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in res.logs.items()}
        res.prog_bar = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in res.prog_bar.items()}
        self.result = res

    def load_batch(self, batch):
        """Copy a new batch into the static input buffers (device-to-device, same shapes)."""
        for k, v in batch.items():
            if k in self.alias_keys:
                continue
            if isinstance(v, torch.Tensor):
                self.static_batch[k].copy_(v, non_blocking=True)
            else:
                self.static_batch[k] = v

    def __call__(self, batch=None):
        if batch is not None:
            self.load_batch(batch)
        self.graph.replay()
        return self.result

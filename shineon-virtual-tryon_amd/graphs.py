"""hipGraph capture of a model's forward + backward (`training_step` + `.backward()`).

One training step of the hot path is ~850 short kernel launches driven from Python; captured once into a
hipGraph (through torch.cuda.CUDAGraph, which records every launch made on the capture stream — including
the ctypes launches into libshineon_hip.so) it replays with a single host call.  The optimizer step and the
data-parallel all-reduce stay outside the graph (two launches, and RCCL is left uncaptured on purpose).

Constraints (same as any stream capture): static shapes, inputs copied into static buffers, no host
synchronisation inside `training_step`; gradients must already be views of the optimizer's flat slab.
"""
import torch

from . import ops

# Thread-local capture: with a process group alive, ProcessGroupNCCL's watchdog thread polls the events of recent collectives
# (hipEventQuery) - under the default GLOBAL capture mode such a call from ANOTHER thread is "not permitted when stream is
# capturing", invalidates the capture and kills the process (seen in about one start in six over a one-rank RCCL group; it
# would hit every multi-GPU run the same way).  Only this thread's own calls need to be capture-safe.
#
# thread_local also relaxes checking for the autograd worker thread (where every backward node of the captured step runs), so
# it is used ONLY while a process group exists; a single process captures in the strict "global" mode.
def capture_mode():
    import torch.distributed as dist

    return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"


class GraphedTrainStep:
    def __init__(self, model, optimizer, sample_batch, warmup=2, alias_keys=(), exchange=None):
        """alias_keys: batch entries used in place (not cloned), e.g. a tensor that is itself the static output of
        another captured graph (the warped cloth handed from the warp stage to the try-on stage)."""
        self.model, self.optimizer = model, optimizer
        self.static_batch = {k: (v.clone() if isinstance(v, torch.Tensor) and k not in alias_keys else v)
                             for k, v in sample_batch.items()}
        self.alias_keys = tuple(alias_keys)
        # only the entries the model reads are refreshed per step (the dataset's batches carry 13 tensors, a model reads ~5;
        # each copy is a launch of its own in front of the graph)
        self.keys = model.batch_keys() if hasattr(model, "batch_keys") else None
        self.result = None
        # trainer.BucketedExchange: its counter bump and "bucket ready" signal kernels are recorded INSIDE the graph, at the
        # points of the backward pass where each gradient bucket is complete
        self.exchange = exchange
        optimizer.zero_grad()  # plants the flat gradient views before anything is recorded
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self._pins = []   # derived-weight copies the graph reads through raw pointers (ops._serve_cached)
        with ops.capture_pins(self._pins), torch.cuda.graph(self.graph, capture_error_mode=capture_mode()):
            self._step()
        self._cache_gen = ops.cache_generation()

    def _step(self):
        self.optimizer.zero_grad()
        if self.exchange is not None:
            self.exchange.begin()
        res = self.model.training_step(self.static_batch, 0)
        ops.backward(res.minimize)
        if self.exchange is not None:
            self.exchange.end()
        # keep only detached values: a live autograd graph would pin the parameters' AccumulateGrad nodes to the
        # stream of an earlier iteration and break the capture that follows
        res.minimize = res.minimize.detach()
        res.logs = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in res.logs.items()}
        res.prog_bar = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in res.prog_bar.items()}
        self.result = res

    def load_batch(self, batch):
        """Copy a new batch into the static input buffers (device-to-device, same shapes): ONE multi-tensor launch."""
        dst, src = [], []
        for k, v in batch.items():
            if k in self.alias_keys:
                continue
            if isinstance(v, torch.Tensor):
                if self.keys is not None and k not in self.keys:
                    continue
                dst.append(self.static_batch[k])
                src.append(v)
            else:
                self.static_batch[k] = v
        _copy_all(dst, src)

    def __call__(self, batch=None):
        _fresh(self)
        if batch is not None:
            self.load_batch(batch)
        self.graph.replay()
        if self.exchange is not None:
            self.exchange.note_replay()
        return self.result


class GraphedChainedStep:
    """The chained warp -> try-on step as THREE hipGraphs replayed on two streams:

        side stream : [warp forward] --ev--> [warp backward] -> (caller: all-reduce, Adam of the warp model) -> next step ...
        main stream :            wait ev ->  copy cloth -> [try-on forward + backward] -> (caller: all-reduce, Adam)

    The warp model's backward pass - many short, low-occupancy kernels - its gradient all-reduce and its optimizer step
    overlap the try-on stage instead of preceding it; the try-on stage only waits for the warped cloth, which it copies
    into its own buffer first, so the side stream may already run the NEXT step's warp stage while this step's try-on
    stage is still busy (same arithmetic, only the order of independent work changes).  The two stages have separate
    static batch buffers and the warp graphs their own scratch slabs (ops.workspace_lane).

    Forward and backward of the warp model are captured separately on one capture stream (autograd runs every backward
    node on the stream of its forward op), the way torch.cuda.make_graphed_callables does."""

    def __init__(self, warp, optw, unet, optu, sample_batch, warmup=2, exchange_w=None, exchange_u=None):
        """exchange_w / exchange_u: trainer.BucketedExchange of the warp / try-on optimizer - their counter bump and "bucket
        ready" signal kernels become nodes of the warp-backward / try-on graph, so each model's gradient buckets can be
        exchanged and applied while the rest of its backward pass is still running."""
        self.warp, self.optw, self.unet, self.optu = warp, optw, unet, optu
        if hasattr(warp, "tower_streams"):
            warp.tower_streams = False   # the warp stage already runs beside the try-on stage: no third stream (warp_model.py)
        self.exchange_w, self.exchange_u = exchange_w, exchange_u
        exw, exu = exchange_w, exchange_u
        clone = lambda: {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in sample_batch.items()}
        self.batch_warp, self.batch_tryon = clone(), clone()
        sb = self.batch_warp
        self.keys_warp = warp.batch_keys() if hasattr(warp, "batch_keys") else None
        self.keys_tryon = (unet.batch_keys() - {"cloth"}) if hasattr(unet, "batch_keys") else None  # cloth comes from the warp stage
        self.side = _side_stream()
        self.fwd_done, self.cloth_taken = torch.cuda.Event(), torch.cuda.Event()
        optw.zero_grad()
        optu.zero_grad()

        def eager():
            with ops.workspace_lane(8):
                optw.zero_grad()
                rw = warp.training_step(sb, 0)
                if exw is not None:
                    exw.begin()
                ops.backward(rw.minimize)
                if exw is not None:
                    exw.end()
            b2 = dict(self.batch_tryon)
            b2["cloth"] = warp.warped_cloth
            optu.zero_grad()
            if exu is not None:
                exu.begin()
            ru = unet.training_step(b2, 0)
            ops.backward(ru.minimize)
            if exu is not None:
                exu.end()

        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

        cs = torch.cuda.Stream()
        self.g_wf, self.g_wb, self.g_u = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with ops.workspace_lane(8):
            with torch.cuda.graph(self.g_wf, stream=cs, capture_error_mode=capture_mode()):
                optw.zero_grad()
                rw = warp.training_step(sb, 0)
            with torch.cuda.graph(self.g_wb, stream=cs, capture_error_mode=capture_mode()):
                if exw is not None:
                    exw.begin()
                ops.backward(rw.minimize)
                if exw is not None:
                    exw.end()
        self.result_warp = _detached(rw)
        del rw
        self.warped = warp.warped_cloth              # static output of the warp-forward graph
        self.cloth_tryon = torch.empty_like(self.warped)  # the try-on stage's private copy (filled by launch_tryon)
        self.cloth_tryon.copy_(self.warped)
        b2 = dict(self.batch_tryon)
        b2["cloth"] = self.cloth_tryon
        self._pins = []   # derived-weight copies (frozen VGG filters) the try-on graph reads through raw pointers
        with ops.capture_pins(self._pins), torch.cuda.graph(self.g_u, capture_error_mode=capture_mode()):
            optu.zero_grad()
            if exu is not None:
                exu.begin()
            ru = unet.training_step(b2, 0)
            ops.backward(ru.minimize)
            if exu is not None:
                exu.end()
        self.result_tryon = _detached(ru)
        del ru
        self._first = True
        self._cache_gen = ops.cache_generation()

    @staticmethod
    def _load(dst, batch, keys=None):
        """Copy a new batch into a stage's static buffers; `keys`: only the entries that stage reads."""
        d, s = [], []
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                if (keys is None or k in keys) and k in dst:
                    d.append(dst[k])
                    s.append(v)
            else:
                dst[k] = v
        _copy_all(d, s)

    def launch_warp_forward(self, batch=None):
        """side stream: (load the new batch,) warp forward.  Only waits for the previous step's try-on stage to have taken
        its copy of the warped cloth (and, by stream order, for the previous warp backward / Adam)."""
        _fresh(self)
        if self._first:
            self.side.wait_stream(torch.cuda.current_stream())
            self._first = False
        else:
            self.side.wait_event(self.cloth_taken)
        with torch.cuda.stream(self.side):
            if batch is not None:
                self._load(self.batch_warp, batch, self.keys_warp)
            self.g_wf.replay()
            self.fwd_done.record(self.side)

    def launch_tryon(self, batch=None):
        """main stream: (load the new batch,) take the warped cloth, then try-on forward + backward."""
        main = torch.cuda.current_stream()
        if batch is not None:
            self._load(self.batch_tryon, batch, self.keys_tryon)
        main.wait_event(self.fwd_done)
        self.cloth_tryon.copy_(self.warped, non_blocking=True)
        self.cloth_taken.record(main)
        self.g_u.replay()
        if self.exchange_u is not None:
            self.exchange_u.note_replay()

    def launch_warp_backward(self):
        """side stream: warp backward; the caller then issues the warp all-reduce / Adam under `with self.on_side():`."""
        with torch.cuda.stream(self.side):
            self.g_wb.replay()
        if self.exchange_w is not None:
            self.exchange_w.note_replay()

    def on_side(self):
        return torch.cuda.stream(self.side)

    def join(self):
        torch.cuda.current_stream().wait_stream(self.side)


def _fresh(step):
    """A captured step reads pinned derived copies of frozen weights (Winograd-domain / transposed VGG filters, ops._serve_cached).
    After ops.invalidate_weight_caches() - load_state_dict, broadcast_parameters: weights overwritten in place - those copies are
    stale, so the step must be captured again."""
    if step._cache_gen != ops.cache_generation():
        raise RuntimeError("parameters were overwritten in place (load_state_dict / broadcast_parameters) after this training step "
                           "was captured; build a new " + type(step).__name__)


def _copy_all(dst, src):
    """dst[i].copy_(src[i]) for same-shape device tensors as one multi-tensor kernel (a copy per batch entry was 5-9 launches
    in front of every replayed step); anything the fused form does not take (other device / dtype / layout) is copied singly."""
    fused = [(d, s) for d, s in zip(dst, src) if s.is_cuda and s.device == d.device and s.dtype == d.dtype and s.shape == d.shape
             and d.is_contiguous() and s.is_contiguous()]
    rest = [(d, s) for d, s in zip(dst, src) if not any(d is f[0] for f in fused)]
    if len(fused) > 1:
        torch._foreach_copy_([d for d, _ in fused], [s for _, s in fused])
    else:
        rest = fused + rest
    for d, s in rest:
        d.copy_(s, non_blocking=True)


def _side_stream():
    """The stream of the warp stage.  (Confining it to a subset of the compute units - hipExtStreamCreateWithCUMask - was
    measured in round 1: 19.1 / 14.5 / 11.9 / 11.2 ms per step with 32 / 64 / 128 / 192 CUs against 8.98 unmasked, because the
    warp forward sits on the next step's critical path; the switch was removed in round 6.)"""
    return torch.cuda.Stream()


def _detached(res):
    res.minimize = res.minimize.detach()
    res.logs = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in res.logs.items()}
    res.prog_bar = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in res.prog_bar.items()}
    return res

"""Lightning-free training / test driver exposing the hooks the reference's Trainer calls
(reference: train.py:32-141; Lightning 0.9 `Trainer(gpus, distributed_backend="ddp", ...)`).

One process per GPU.  Data parallelism = mean all-reduce of the flat gradient slab over RCCL (torch.distributed
backend "nccl"), a few large buckets, issued asynchronously so the transfer over xGMI overlaps whatever the step still
has to do; BatchNorm statistics stay per rank like the reference's nn.BatchNorm2d under DDP, with rank 0's running
statistics broadcast once per step as ONE flat tensor (DDP's broadcast_buffers).

The step itself lives in two small engines that bench.py and Trainer.fit share:
  TrainStep         one model: forward+backward as a replayed hipGraph (graph=True), asynchronous gradient exchange,
                    fused Adam; the exchange + Adam of step i are completed right before step i+1's graph is launched
                    (overlap=True), i.e. they overlap the host-side batch preparation of step i+1.
  ChainedTrainStep  the warp -> try-on pair of SURVEY 8d C4: three hipGraphs on two streams (graphs.GraphedChainedStep) or two
                    sequential graphs, both gradient exchanges hidden behind the other model's compute.
"""
import ctypes
import logging
import os
import os.path as osp
import signal
import time

import torch
import torch.distributed as dist

from . import ops
from .options import str2num

logger = logging.getLogger("logger")


def init_distributed(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* if launched under torchrun."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not dist.is_initialized() and _single_rank_group() and torch.cuda.is_available():
        # SHINEON_SINGLE_RANK_GROUP=1: a ONE-rank RCCL group, so that a one-GPU box runs the whole exchange path for real -
        # communicator set-up, per-bucket all-reduce on the communication stream behind the in-graph signals, broadcasts
        import socket

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        torch.cuda.set_device(int(os.environ.get("SHINEON_LOCAL_DEVICE", "0")))
        dist.init_process_group(backend=backend or "nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                **_group_timeout())
        return 0, 1
    if world <= 1 or dist.is_initialized():
        return dist.get_rank() if dist.is_initialized() else 0, max(world, 1)
    # SHINEON_DIST_BACKEND=gloo: functional runs of the multi-rank code path where RCCL cannot be used (e.g. two ranks
    # sharing ONE GPU in tests/test_00_multi_rank_gpu.py); SHINEON_LOCAL_DEVICE pins the device index independently of LOCAL_RANK
    backend = backend or os.environ.get("SHINEON_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("SHINEON_LOCAL_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    dist.init_process_group(backend=backend, **_group_timeout())
    return dist.get_rank(), dist.get_world_size()


def _group_timeout():
    """SHINEON_DIST_TIMEOUT_S: fail a broken rendezvous / a collective nobody joins in minutes, not in the 10-30 minute default."""
    if not os.environ.get("SHINEON_DIST_TIMEOUT_S"):
        return {}
    import datetime

    return {"timeout": datetime.timedelta(seconds=float(os.environ["SHINEON_DIST_TIMEOUT_S"]))}


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _single_rank_group():
    return os.environ.get("SHINEON_SINGLE_RANK_GROUP", "0") == "1"


def _collective():
    """True when gradients and buffers have to travel: more than one rank - or the one-rank RCCL group of
    SHINEON_SINGLE_RANK_GROUP=1 (init_distributed), where every collective is issued although nothing changes hands."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _single_rank_group())


class GradientAllReducer:
    """Mean all-reduce of a flat gradient slab in `n_buckets` contiguous pieces.

    xGMI is point-to-point (7 links per GPU), so a few large messages are preferred to DDP's many 25 MB
    buckets: the 76-91 MB slab is cut into 4 pieces by default."""

    def __init__(self, flat_grads, n_buckets=4, group=None):
        self.flat = flat_grads
        self.group = group
        n_buckets = max(1, int(n_buckets))
        n = flat_grads.numel()
        step = (n + n_buckets - 1) // n_buckets
        step = (step + 1023) // 1024 * 1024
        self.buckets = [flat_grads[i:min(i + step, n)] for i in range(0, n, step)]
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = _collective()   # False: no collective is issued (one process; bench.py also clears it to time a step without)
        self._works = None

    def start(self):
        """Issue the asynchronous all-reduce of every bucket (RCCL runs on its own stream, ordered after the work
        already queued on the current stream), so that it overlaps whatever is launched next."""
        if self.active:
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
        if not self.active:
            return 1.0 / self.world
        self.start()
        return self.finish()


class BucketedExchange:
    """Gradient exchange OVERLAPPED WITH THE BACKWARD PASS of a single model (what torch DDP does for the reference,
    train.py:52-57,76-85), also when forward + backward replay as one hipGraph.

    The optimizer's flat gradient slab is cut at parameter boundaries into buckets of ~`bucket_bytes` (xGMI is point-to-point:
    ring collectives are per-link bound, so buckets are sized in bytes - 64 MB by default - not a fixed count).  The backward
    pass produces the slab from its END (last layers first).  While it is being issued - eagerly or under stream capture -
    every operator reports the parameters whose gradient kernels it has just launched (ops.grad_ready, plus autograd's
    post-accumulate hooks for gradients that go through AccumulateGrad); when all parameters of a bucket have reported as
    often as in the calibration pass, a one-thread kernel stores the step counter into that bucket's signal word
    (so_signal_store: a node of the graph when capturing).  After launching the step the host queues, per bucket in readiness
    order, on the communication stream:  hipStreamWaitValue32(word >= step)  ->  all-reduce of the bucket  ->  Adam on the
    bucket's slab range.  The exchange and the update of the last layers thus run while the backward pass of the earlier
    layers is still executing; nothing here synchronises the host.

    Safety: a bucket's parameters are read for the last time by the input-gradient kernels launched before its signal node
    (same stream), its gradients are written for the last time there too; buckets are disjoint slab ranges.  finish() makes
    the compute stream wait for the communication stream before the next forward.

    Invariants that are CHECKED, not assumed:
      * every pass reports each parameter exactly as often as the calibration pass did (end(): a parameter reporting MORE
        often would have released its bucket in mid-backward - partial gradients reduced, weights rewritten under dgrad
        kernels that still read them; under capture that would be baked into every replay).  A mismatch raises; call
        recalibrate() when the model's pass structure changes on purpose;
      * launch() is only legal after the step holding the matching signal nodes has been enqueued (begin() or note_replay()
        since the previous launch): the default wait is a polling kernel on a stream that shares hardware queues with the
        compute streams - queued AHEAD of its producer it would spin with nothing behind it;
      * the polling wait is bounded (so_stream_wait_ge_bounded): abort() - called by the engines on any exception path - or
        the deadline (SHINEON_WAIT_DEADLINE_S, default 60 s) makes it return with a status the next finish() raises on."""

    def __init__(self, optimizer, bucket_bytes=64 << 20, group=None):
        from ._lib import check, lib

        self.opt, self.group = optimizer, group
        # False (round 6): ONE Adam pass behind the last bucket's all-reduce.  True = the update of each bucket right behind its
        # all-reduce, i.e. beside the rest of the backward pass: measured over the one-rank RCCL group on c3 the HBM-bound update
        # slows the backward kernels by what it saves (786.8 / 788.2 vs 793.7 / 789.5 frames/s, profiles/r06_single_rank_rccl.txt)
        self.adam_per_bucket = False
        self.L, self._check = lib(), check
        if not self.L.so_signal_can_wait():
            raise RuntimeError("this device does not support hipStreamWaitValue32; use GradientAllReducer")
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = _collective()
        table = optimizer.slot_table()
        total = optimizer.flat_grads.numel()
        want = max(1, int(bucket_bytes) // 4)
        n_min = 2     # at least two buckets, so that something can overlap
        n_b = max(n_min, min(len(table), (total + want - 1) // want))
        per = (total + n_b - 1) // n_b
        self.buckets = []   # [lo, hi, parameters] in slab order
        lo, members = 0, []
        for p, off, size in table:
            members.append(p)
            if off + size - lo >= per and len(self.buckets) < n_b - 1:
                self.buckets.append([lo, off + size, members])
                lo, members = off + size, []
        if members:
            self.buckets.append([lo, total, members])
        self.bucket_of = {id(p): b for b, (_, _, ps) in enumerate(self.buckets) for p in ps}
        self.flags = [self.L.so_signal_alloc() for _ in self.buckets]
        if not all(self.flags):
            raise RuntimeError("hipExtMallocWithFlags(hipMallocSignalMemory) failed")
        self.counter = torch.zeros(2, dtype=torch.int32, device=optimizer.flat_grads.device)
        self.comm = torch.cuda.Stream()
        self.expected = None        # {id(param): reports per backward pass}, from the calibration pass
        self.order = None           # bucket indices in the order they became ready in the calibration pass
        self._count, self._signalled, self._calib_order = {}, set(), []
        self._streams = {}          # bucket -> {stream handle: torch stream} the bucket's gradient kernels were issued on
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_ready) for p, _, _ in table]
        self.step_no = 0
        self._launched = False
        self._armed = False
        self._launched_step = 0     # step_no of the last launch(): launch() must follow a newly enqueued step
        self.failed = False         # set when the owning engine's __call__ raised: no further launch()
        # [abort request, waiter status]: pinned host words the polling kernels read / write (include/shineon_hip.h)
        self.words = self.L.so_hostwords_alloc()
        if not self.words:
            raise RuntimeError("hipHostMalloc of the abort / status words failed")
        self._words = (ctypes.c_uint32 * 2).from_address(self.words)
        self.deadline_ticks = int(float(os.environ.get("SHINEON_WAIT_DEADLINE_S", "60")) * 1e8)   # 100 MHz wall clock
        # the waiter is a one-lane polling kernel (so_stream_wait_ge_bounded).  hipStreamWaitValue32 was measured and removed:
        # the command-processor wait slows the dispatch of every kernel of the step it waits through (6.64 vs 5.87 ms/step, c3)

    def __del__(self):
        try:
            for f in self.flags:
                self.L.so_signal_free(f)
            if getattr(self, "words", 0):
                self._words = None
                self.L.so_hostwords_free(self.words)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def abort(self):
        """Emergency release of every polling wait that is queued or running (they return within ~64 polls), for a caller that
        KNOWS a bucket's signal node will never run.  The all-reduce + Adam behind a released wait then run on whatever the
        gradient slab holds (check_status() / exchange_statuses() report it); the exchange is unusable afterwards.  The step
        engines do not call this on ordinary exceptions (see _abort_exchanges_on_error)."""
        if self._words is not None:
            self._words[0] = 1

    def check_status(self):
        code = int(self._words[1]) if self._words is not None else 0
        if code:
            raise RuntimeError("gradient-bucket wait " + {1: "was aborted", 2: "passed its deadline (SHINEON_WAIT_DEADLINE_S): a bucket's "
                               "signal node never ran"}.get(code, f"failed with status {code}") + "; the step's gradients are incomplete")

    def recalibrate(self):
        """The model's pass structure changed on purpose (another number of generator passes, a weight now shared): the next
        begin() / end() pair is a calibration pass again (buckets released at end())."""
        self.expected = self.order = None

    # ---- while the backward pass is being issued ---------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def begin(self):
        """Call right before forward + backward are issued (inside the capture when capturing): bumps the device-side step
        counter and arms the readiness hooks."""
        from . import ops

        self._check(self.L.so_counter_bump(self.counter.data_ptr(), self._stream()), "counter_bump")
        if not torch.cuda.is_current_stream_capturing():
            self.step_no += 1           # the device counter advances when the bump EXECUTES: now, or at every graph replay
        self._count, self._signalled, self._calib_order = {}, set(), []
        self._streams = {}
        ops._GRAD_READY[0] = self._on_ready
        self._armed = True

    def _on_ready(self, p):
        b = self.bucket_of.get(id(p))
        if b is None or not self._armed:   # (the autograd hooks also fire in backward passes this engine does not drive)
            return
        self._count[id(p)] = self._count.get(id(p), 0) + 1
        cur = torch.cuda.current_stream()
        self._streams.setdefault(b, {})[cur.cuda_stream] = cur   # the reporting operator's kernels are in THIS stream
        if self.expected is None or b in self._signalled:
            if self.expected is None and (not self._calib_order or self._calib_order[-1] != b):
                self._calib_order.append(b)
            return
        if all(self._count.get(id(q), 0) >= self.expected.get(id(q), 0) for q in self.buckets[b][2]):
            self._signal(b)

    def _signal(self, b):
        # A bucket may hold gradients produced on several streams (WarpModel.tower_streams runs the cloth tower - forward and,
        # through autograd, backward - on a forked stream; a bucket is a contiguous slab range and can straddle both towers).
        # The signal is a kernel on ONE stream: it first waits for everything the other reporting streams have been given so
        # far (all of the bucket's reports are in by now, each made after its kernels were launched), eagerly and as graph
        # edges under capture.  Without this the communication stream could reduce and update a bucket whose other tower's
        # weight-gradient kernels are still writing it.
        cur = torch.cuda.current_stream()
        for handle, st in self._streams.get(b, {}).items():
            if handle != cur.cuda_stream:
                ev = torch.cuda.Event()
                ev.record(st)
                cur.wait_event(ev)
        self._check(self.L.so_signal_store(self.flags[b], self.counter.data_ptr(), 0, self._stream()),
                    "signal_store")
        self._signalled.add(b)

    def end(self):
        """Call right after backward has been issued: buckets that did not complete through the hooks (parameters without a
        gradient this step) are signalled here, i.e. at the end of the backward pass."""
        from . import ops

        ops._GRAD_READY[0] = None
        self._armed = False
        if self.expected is not None and self._count != self.expected:
            # (eager steps, warm-up passes and the capture pass alike: nothing has been launched on the communication stream
            #  for this step yet, and a capture that raises here is discarded)
            names = {id(p): i for i, (p, _, _) in enumerate(self.opt.slot_table())}
            diff = [(names.get(k, -1), self.expected.get(k, 0), self._count.get(k, 0))
                    for k in set(self.expected) | set(self._count) if self.expected.get(k, 0) != self._count.get(k, 0)]
            raise RuntimeError(f"BucketedExchange: {len(diff)} parameters reported their gradient another number of times than in "
                               f"the calibration pass (slot, expected, seen): {sorted(diff)[:8]} - a bucket may have been released "
                               "before its gradients were complete; call recalibrate() if the pass structure changed on purpose")
        if self.expected is None:       # calibration pass: remember how often each parameter reports and the bucket order
            self.expected = dict(self._count)
            seen, order = set(), []
            for b in reversed(self._calib_order):   # a bucket is ready at its LAST report
                if b not in seen:
                    seen.add(b)
                    order.append(b)
            order.reverse()
            self.order = order + [b for b in range(len(self.buckets)) if b not in seen]
        for b in range(len(self.buckets)):
            if b not in self._signalled:
                self._signal(b)

    def note_replay(self):
        """A captured step (with begin() / end() recorded inside) has just been replayed."""
        self.step_no += 1

    # ---- after the step has been launched ----------------------------------------------------------------------------
    def launch(self, grad_scale_extra=1.0):
        """Queue wait -> all-reduce -> Adam per bucket on the communication stream.  Non-blocking for the host with RCCL."""
        if self.failed:
            raise RuntimeError("BucketedExchange: the step engine raised earlier; rebuild the engine")
        if self.step_no <= self._launched_step:
            raise RuntimeError("BucketedExchange.launch() without a newly enqueued step (begin() / note_replay() since the last "
                               "launch): the polling wait would be queued ahead of the kernels that release it")
        self.check_status()
        self._launched_step = self.step_no
        self.opt.begin_step()
        scale = grad_scale_extra / self.world
        flat = self.opt.flat_grads
        for b in self.order:
            lo, hi, _ = self.buckets[b]
            self._check(self.L.so_stream_wait_ge_bounded(self.flags[b], self.step_no, self.words, self.deadline_ticks,
                                                         self.comm.cuda_stream), "stream_wait_ge_bounded")
            with torch.cuda.stream(self.comm):
                if self.active:
                    dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
                if self.adam_per_bucket:
                    self.opt.step_range(lo, hi, grad_scale=scale)
        if not self.adam_per_bucket:
            # ONE Adam pass over the whole slab behind the last bucket's all-reduce: element-wise, so bit-identical to the
            # per-bucket form - which ran its HBM-bound update beside the backward pass and slowed that by what it saved
            with torch.cuda.stream(self.comm):
                self.opt.step_range(0, flat.numel(), grad_scale=scale)
        self._launched = True

    def finish(self):
        """The compute stream waits for every bucket's update (call before the next forward / a checkpoint)."""
        if self._launched:
            torch.cuda.current_stream().wait_stream(self.comm)
            self._launched = False
        self.check_status()   # host-pinned word: no synchronisation; a time-out of step k surfaces at the latest one step later

    def describe(self):
        sizes = [(hi - lo) * 4 / 1e6 for lo, hi, _ in self.buckets]
        return f"{len(self.buckets)} buckets of " + "/".join(f"{x:.0f}" for x in sizes) + " MB, exchange overlapped with backward"


def _abort_exchanges_on_error(method):
    """An exception escaping a step engine's __call__ marks its exchanges failed (no further launch()) and propagates; the
    process is expected to exit non-zero (Trainer.fit saves its interrupt checkpoint first).

    It does NOT set the abort word: every bounded wait that launch() has queued belongs to a step whose signal nodes were
    all enqueued before it (launch() refuses anything else), so those waits are released by the device on their own and
    the all-reduce + Adam behind them run on COMPLETE gradients.  Aborting them - what this wrapper did before - released
    the previous, correctly enqueued step early whenever a host-side error hit the next one (bad batch, a failing buffer
    broadcast) and let Adam run on partial gradients right before the interrupt checkpoint was written.  A wait whose
    signal really never runs (a graph launch that failed asynchronously) ends at its wall-clock deadline with status 2;
    BucketedExchange.abort() stays available for a caller that knows the producer will never run."""
    import functools

    @functools.wraps(method)
    def wrapped(self, *a, **kw):
        try:
            return method(self, *a, **kw)
        except BaseException:
            for ex in self._all_exchanges():
                if ex is not None:
                    ex.failed = True
            raise

    return wrapped


def exchange_statuses(step):
    """Status words of every bucketed exchange of a step engine after the device has been synchronised: {} when all waits
    completed normally, else {exchange index: 1 (aborted) | 2 (deadline)} - the parameters / moments then hold an update made
    from incomplete gradients."""
    out = {}
    for i, ex in enumerate(getattr(step, "_all_exchanges", lambda: ())()):
        code = int(ex._words[1]) if ex is not None and ex._words is not None else 0
        if code:
            out[i] = code
    return out


def _make_exchange(optimizer, bucketed, bucket_bytes, group=None):
    env = os.environ.get("SHINEON_BUCKETED")
    use = _collective() if bucketed is None else bool(bucketed)
    if env is not None and bucketed is None:
        use = env == "1"
    if not use:
        return None
    if os.environ.get("SHINEON_BUCKET_MB"):
        bucket_bytes = int(float(os.environ["SHINEON_BUCKET_MB"]) * (1 << 20))
    try:
        return BucketedExchange(optimizer, bucket_bytes, group=group)
    except RuntimeError as e:   # no hipStreamWaitValue32 on this device: the whole-slab exchange after the graph
        logger.warning("bucketed gradient exchange unavailable (%s); falling back to GradientAllReducer", e)
        return None


# ---- buffers / parameters: ONE collective each ------------------------------------------------------------------------
def flatten_float_buffers(model):
    """Re-home every floating-point buffer of `model` (BatchNorm running_mean / running_var) as a view of ONE flat tensor
    and return it (None if there are none).  state_dict keys, shapes and values are unchanged; kernels keep writing the
    running statistics through the views.  Idempotent while the buffers have not been moved (model.to())."""
    cached = getattr(model, "_so_flat_buffers", None)
    entries = [(m, name, b) for m in model.modules() for name, b in m._buffers.items()
               if b is not None and b.is_floating_point()]
    if not entries:
        return None
    if cached is not None and all(b.device == cached.device and b.data_ptr() >= cached.data_ptr() and
                                  b.data_ptr() < cached.data_ptr() + cached.numel() * 4 for _, _, b in entries):
        return cached
    sizes = [(b.numel() + 3) // 4 * 4 for _, _, b in entries]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=entries[0][2].device)
    off = 0
    for (m, name, b), sz in zip(entries, sizes):
        view = flat[off:off + b.numel()].view(b.shape)
        view.copy_(b)
        m._buffers[name] = view
        off += sz
    object.__setattr__(model, "_so_flat_buffers", flat)
    return flat


def broadcast_buffers(model, src=0, async_op=False):
    """DDP's broadcast_buffers=True: rank 0's BatchNorm running statistics win - one collective over the flat buffer
    tensor.  (num_batches_tracked is advanced identically on every rank, so the integer counters need no exchange.)"""
    if not _collective():
        return None
    flat = flatten_float_buffers(model)
    if flat is None:
        return None
    return dist.broadcast(flat, src, async_op=async_op)


def broadcast_parameters(model, src=0, optimizer=None):
    """Rank 0's initial weights everywhere (DDP's constructor broadcast).  With the optimizer given, the whole flat
    parameter slab travels as one message; otherwise one message per parameter."""
    if not _collective():
        return
    if optimizer is not None:
        dist.broadcast(optimizer.flat_params, src)
    else:
        for p in model.parameters():
            dist.broadcast(p.data, src)
    ops.invalidate_weight_caches()   # the broadcast writes through .data: no version bump for the derived-weight caches to see
    broadcast_buffers(model, src)


class _BufferSnapshot:
    """Engine construction runs the model a few times without an optimizer step (plan measurement, graph warm-up): those
    passes must not leave a trace in BatchNorm's running statistics / batch counters.  Values are put back IN PLACE (the
    buffers may be views of flat tensors that kernels and graphs hold by address)."""

    def __init__(self, *models):
        for m in models:  # buffers get their final home first (flat float tensor, shared BatchNorm counters)
            flatten_float_buffers(m)
            if hasattr(m, "plant_shared_buffers"):
                m.plant_shared_buffers()
        self.saved = [(b, b.detach().clone()) for m in models for b in m.buffers()]

    def restore(self):
        with torch.no_grad():
            for b, v in self.saved:
                b.copy_(v)


def _to_device(batch, device):
    return {k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


# ---- step engines ------------------------------------------------------------------------------------------------------
class TrainStep:
    """One optimisation step of one model: (load batch ->) forward + backward -> gradient exchange -> Adam."""

    def __init__(self, model, optimizer, sample_batch, graph=True, overlap=True, accumulate=1, sync_buffers=True,
                 bucketed=None, bucket_bytes=64 << 20):
        """bucketed: exchange + update per gradient bucket while the backward pass is still running (BucketedExchange);
        None = when there is more than one rank (SHINEON_BUCKETED=1 / 0 forces it on / off, e.g. single-rank tests)."""
        self.model, self.optimizer = model, optimizer
        self.accumulate = max(1, int(accumulate))
        self.graph = bool(graph) and self.accumulate == 1   # gradient accumulation re-enters backward: eager only
        self.overlap = bool(overlap)
        self.sync_buffers = sync_buffers and _collective() and flatten_float_buffers(model) is not None
        optimizer.zero_grad()
        self.reducer = GradientAllReducer(optimizer.flat_grads)
        self.exchange = _make_exchange(optimizer, bucketed, bucket_bytes) if self.accumulate == 1 else None
        self._pending = False
        self._micro = 0
        self._graphed = None
        self._shapes = {k: tuple(v.shape) for k, v in sample_batch.items() if isinstance(v, torch.Tensor)}
        if self.graph:
            from .graphs import GraphedTrainStep

            snap = _BufferSnapshot(model)
            self._eager(sample_batch, update=False)  # measures igemm plans, sizes every scratch slab before the capture
            self._graphed = GraphedTrainStep(model, optimizer, sample_batch, exchange=self.exchange)
            torch.cuda.synchronize()
            snap.restore()

    def _fits(self, batch):
        return all(tuple(batch[k].shape) == s for k, s in self._shapes.items() if k in batch)

    def _eager(self, batch, update=True):
        if self._micro == 0:
            self.optimizer.zero_grad()
        if self.exchange is not None:
            self.exchange.begin()
        res = self.model.training_step(batch, 0)
        ops.backward(res.minimize / self.accumulate if self.accumulate > 1 else res.minimize)
        if self.exchange is not None:
            self.exchange.end()
        self._micro += 1
        if update and self._micro == self.accumulate:
            self._micro = 0
            self._start_exchange()
            self._pending = True
            if not self.overlap:
                self.flush()
        elif not update:
            self._micro = 0
        return res

    def _start_exchange(self):
        if self.exchange is not None:
            self.exchange.launch()      # per bucket on the communication stream: wait for its signal, all-reduce, Adam
        else:
            self.reducer.start()

    def flush(self):
        """Complete the outstanding gradient exchange and apply Adam (no-op if nothing is pending)."""
        if self._pending:
            if self.exchange is not None:
                self.exchange.finish()
            else:
                self.optimizer.step(grad_scale=self.reducer.finish())
            self._pending = False

    @property
    def stepped(self):
        """True when the last call completed an accumulation window (an optimizer step was issued or is pending)."""
        return self._micro == 0

    def _all_exchanges(self):
        return (self.exchange,)

    @_abort_exchanges_on_error
    def __call__(self, batch):
        self.flush()  # parameters of the previous step must have landed before this forward
        if self.sync_buffers:
            broadcast_buffers(self.model)
        if self._graphed is not None and self._fits(batch):
            res = self._graphed(batch)
            self._start_exchange()
            self._pending = True
            if not self.overlap:
                self.flush()
            return res
        return self._eager(batch)  # eager mode, or a ragged last batch of the epoch


class ChainedTrainStep:
    """The chained warp -> try-on step (the reference hands the warped cloth over through PNG files,
    models/warp_model.py:143-149 -> datasets/vvt_dataset.py:139-150; in process it is the detached tensor):
    WarpModel fwd+bwd+Adam, then UnetMaskModel fwd+bwd+Adam on the warped cloth.

    schedule = "pipeline": three hipGraphs on two streams (graphs.GraphedChainedStep) - side stream: warp forward ->
        warp backward -> warp gradient exchange -> warp Adam; main stream: try-on fwd+bwd -> its exchange (hidden behind
        the next step's warp forward) -> try-on Adam.
    schedule = "sequential": two graphs back to back; the warp exchange hides behind the try-on graph, the try-on
        exchange behind the next warp graph.
    schedule = "eager": every kernel launched from Python (profiling / PMC passes).
    schedule = "auto": "pipeline" on one rank; with several ranks the try-on exchange is timed once on this node and
        "pipeline" is kept only if it is cheaper than what the two-stream schedule gains (`pipeline_gain_ms`)."""

    def __init__(self, warp, optw, unet, optu, sample_batch, schedule="auto", pipeline_gain_ms=None, sync_buffers=True, log=None,
                 bucketed=None, bucket_bytes=64 << 20):
        """pipeline_gain_ms: what the two-stream schedule saves per step on ONE rank; None = measure it here (both schedules
        are built and replayed a few times without optimizer steps, MAX over ranks) when the choice has to be made.
        bucketed: each model's gradient exchange + Adam per bucket on its own communication stream, released by signal nodes
        inside that model's captured backward pass (BucketedExchange, as in trainer.TrainStep); None = whenever collectives
        are issued (SHINEON_BUCKETED=0 / 1 forces it).  Measured on MI355X over a ONE-rank RCCL group
        (SHINEON_SINGLE_RANK_GROUP=1: every collective issued, nothing on the wire; 6.66 ms/step without collectives;
        profiles/r03_single_rank_rccl.txt):
          one communicator for everything   whole-slab exchange after each graph 7.28 ms (+0.62, whatever the bucket count),
                                            bucketed 8.39 ms - the warp model's buffer broadcast of step k+1 queues behind
                                            the try-on all-reduce of step k on RCCL's stream, i.e. behind the END of the
                                            try-on graph, and the two-stream overlap is lost
          try-on exchange on its own        whole-slab 6.87 ms (+0.22), bucketed 6.73 ms (+0.07) - removed in round 6
          communicator                      (deadlock-prone with real peers; the default below needs no second one)
        Also tried: collectives issued on the graph streams themselves (8.31 ms); GPU_MAX_HW_QUEUES=8 (doubles the step time
        with or without collectives)."""
        self.warp, self.optw, self.unet, self.optu = warp, optw, unet, optu
        self.batch = sample_batch
        if hasattr(warp, "tower_streams"):
            warp.tower_streams = False   # a third concurrent stream slows the chained schedules down (warp_model.py)
        optw.zero_grad()
        optu.zero_grad()
        # ONE communicator by default: both models' collectives are issued in the same host order on every rank and run in
        # that order on RCCL's stream, so no rank can start them in another order than its peers.  (Two communicators whose
        # kernels are released by device-side events start in a RANK-DEPENDENT order; when their streams share a hardware
        # queue each rank can end up waiting inside one collective for a peer that sits inside the other - the pattern NCCL
        # documents as deadlock-prone.  It has only ever run over the ONE-rank group here, where nothing waits for a peer.)
        # What made one communicator slow in round 3 (+0.62 ms/step) was the warp model's per-step buffer broadcast queueing
        # behind the try-on all-reduce, i.e. behind the end of the try-on graph; the broadcast is now LAZY (see
        # `lazy_buffers`), which removes that collective from the step altogether.
        # (Round 3's second communicator for the try-on exchange - 0.1 ms less exposed over the one-rank group - was removed in
        # round 6: it never met a real wire and is the deadlock-prone arrangement.)
        self.redw, self.redu = GradientAllReducer(optw.flat_grads), GradientAllReducer(optu.flat_grads)
        self.exw = self.exu = None
        if bucketed is None and os.environ.get("SHINEON_BUCKETED") is None:
            bucketed = False   # whole-slab exchange after each graph: host-ordered on the one communicator
        if schedule != "eager":
            self.exw = _make_exchange(optw, bucketed, bucket_bytes)
            self.exu = _make_exchange(optu, bucketed, bucket_bytes) if self.exw is not None else None
        self.sync_buffers = sync_buffers and _collective() and flatten_float_buffers(warp) is not None
        # BatchNorm running statistics are not READ by a training-mode forward, so DDP's per-forward broadcast of rank 0's
        # buffers only matters when somebody looks at them: validation, a checkpoint, the end of training.  Lazy = rank 0's
        # statistics are broadcast in synchronize() (every consumer calls it first) instead of before every step - the
        # values every observer sees are the ones DDP would show (rank 0's chain of updates), one collective per step less.
        # SHINEON_BROADCAST_BUFFERS_EVERY_STEP=1 restores the literal per-step broadcast.
        self.lazy_buffers = os.environ.get("SHINEON_BROADCAST_BUFFERS_EVERY_STEP", "0") != "1"
        self._pending_u = False
        self._u_started = False      # whole-slab mode: the try-on all-reduce of the pending step has been issued
        self._gp = self._gw = self._gu = None
        self.exchange_ms = self.pipeline_gain_ms = None
        auto = schedule == "auto" and _collective()
        if schedule == "auto":
            schedule = "pipeline"
        self.schedule = schedule
        snap = _BufferSnapshot(warp, unet)
        if schedule != "eager":
            self.eager_step(sample_batch, update=False)  # plans measured, scratch slabs sized, gradient views planted
        if auto:
            # with several ranks the try-on exchange has no other graph to hide behind in the two-stream schedule: keep that
            # schedule only if what it gains on this node (measured, not assumed) exceeds what the exchange costs (measured)
            self.exchange_ms = self._time_exchange(optu.flat_grads)
            if self.exu is not None:
                # bucketed: every bucket but the one completed last (the first layers') travels while the try-on backward pass
                # is still running; what stays exposed is about one bucket's share of the whole-slab time (an estimate)
                self.exchange_ms /= len(self.exu.buckets)
            self._build("pipeline", sample_batch)
            t_pipe = self._time_graphs()
            if pipeline_gain_ms is None:
                self._build("sequential", sample_batch)
                t_seq = self._time_graphs(sequential=True)
                pipeline_gain_ms = self._max_over_ranks(t_seq - t_pipe)
            self.pipeline_gain_ms = pipeline_gain_ms
            self.schedule = "pipeline" if self.exchange_ms < pipeline_gain_ms else "sequential"
            if self.schedule == "sequential":
                self._gp = None
                if self._gw is None:
                    self._build("sequential", sample_batch)
            else:
                self._gw = self._gu = None
            if log:
                log(f"try-on gradient all-reduce ({optu.flat_grads.numel() * 4 / 1e6:.1f} MB, {_world()} ranks): "
                    f"{self.exchange_ms:.2f} ms exposed in the two-stream schedule, which gains {pipeline_gain_ms:.2f} ms/step "
                    f"on one rank -> {self.schedule} schedule")
        elif schedule in ("pipeline", "sequential"):
            self._build(schedule, sample_batch)
        elif schedule != "eager":
            raise ValueError(f"unknown schedule {schedule!r}")
        torch.cuda.synchronize()
        snap.restore()

    def _build(self, schedule, sample_batch):
        if schedule == "pipeline":
            from .graphs import GraphedChainedStep

            self._gp = GraphedChainedStep(self.warp, self.optw, self.unet, self.optu, sample_batch, exchange_w=self.exw,
                                          exchange_u=self.exu)
        else:
            from .graphs import GraphedTrainStep

            self._gw = GraphedTrainStep(self.warp, self.optw, sample_batch, exchange=self.exw)
            b2 = dict(sample_batch)
            b2["cloth"] = self.warp.warped_cloth.detach()  # static output of the warp graph, consumed in place
            self._gu = GraphedTrainStep(self.unet, self.optu, b2, alias_keys=("cloth",), exchange=self.exu)

    @staticmethod
    def _max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        if _collective():
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def _time_graphs(self, sequential=False, reps=5):
        """ms per step of the captured graphs alone (no exchange, no optimizer step: the gradients are simply overwritten)."""
        def once():
            if sequential:
                self._gw()
                self._gu()
            else:
                self._gp.launch_warp_forward()
                self._gp.launch_tryon()
                self._gp.launch_warp_backward()

        for _ in range(2):
            once()
        if not sequential:
            self._gp.join()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            once()
        if not sequential:
            self._gp.join()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    @staticmethod
    def _time_exchange(flat, reps=5):
        """All-reduce time of a slab of this size on this node (scratch copy; MAX over ranks so all take one decision)."""
        scratch = torch.zeros_like(flat)
        probe = GradientAllReducer(scratch)
        for _ in range(2):
            probe.all_reduce()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            probe.all_reduce()
        e1.record()
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / reps], dtype=torch.float64, device=flat.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    @property
    def launch_description(self):
        return {"eager": "eager",
                "sequential": "two sequential hipGraphs (warp, try-on); Adam and all-reduce eager",
                "pipeline": "three hipGraphs on two streams: warp forward -> [try-on fwd+bwd || warp backward + its "
                            "all-reduce + Adam]; Adam and all-reduce eager"}[self.schedule]

    def eager_step(self, batch=None, update=True):
        batch = self.batch if batch is None else batch
        self.flush()
        if self.sync_buffers and update and not self.lazy_buffers:
            broadcast_buffers(self.warp)
        self.optw.zero_grad()
        rw = self.warp.training_step(batch, 0)
        ops.backward(rw.minimize)
        if update:
            self.optw.step(grad_scale=self.redw.all_reduce())
        b2 = dict(batch)
        b2["cloth"] = self.warp.warped_cloth.detach()
        self.optu.zero_grad()
        ru = self.unet.training_step(b2, 0)
        ops.backward(ru.minimize)
        if update:
            self.optu.step(grad_scale=self.redu.all_reduce())
        return rw, ru

    def flush(self):
        """Land the try-on update that is still travelling (its exchange overlaps the next step's warp forward)."""
        if self._pending_u:
            if self.exu is not None:
                self.exu.finish()
            else:
                if not self._u_started:
                    self.redu.start()
                self.optu.step(grad_scale=self.redu.finish())
            self._pending_u = False
            self._u_started = False

    def _all_exchanges(self):
        return (self.exw, self.exu)

    @_abort_exchanges_on_error
    def __call__(self, batch=None):
        """One chained step; `batch` (device tensors of the captured shapes) is copied into the static buffers, None
        re-uses the resident batch.  Results are device tensors valid after the streams are synchronised."""
        if self.schedule == "eager":
            return self.eager_step(batch)
        gp = self._gp
        if gp is not None:
            if self.sync_buffers and not self.lazy_buffers:
                with gp.on_side():
                    broadcast_buffers(self.warp)
            if self.exw is not None:
                with gp.on_side():
                    self.exw.finish()   # the previous step's warp update has landed before this warp forward
            if self.exw is None and self.exu is None:
                # Whole-slab exchange on ONE communicator: collectives run in host issue order, so they are issued in the order
                # they become READY on the device.  The warp stage of step k (forward, backward - neither depends on the
                # try-on stage) runs entirely beside the try-on stage of step k-1, so its gradients W_k are ready BEFORE the
                # try-on gradients U_{k-1}: W_k is issued first, U_{k-1} behind it.  (Issued the other way round - round 3's
                # order - W_k queues behind a collective that cannot start before the try-on graph ends: 0.64 ms per step
                # over the one-rank RCCL group, profiles/r04_single_rank_rccl.txt.)
                gp.launch_warp_forward(batch)
                gp.launch_warp_backward()
                with gp.on_side():
                    self.redw.start()
                if self._pending_u and not self._u_started:
                    self.redu.start()           # U_{k-1}: waits for the try-on graph launched by the previous call
                    self._u_started = True
                with gp.on_side():
                    self.optw.step(grad_scale=self.redw.finish())
                self.flush()                    # try-on Adam of step k-1 on the main stream
                gp.launch_tryon(batch)
                self._pending_u, self._u_started = True, False
                return gp.result_warp, gp.result_tryon
            gp.launch_warp_forward(batch)
            self.flush()
            gp.launch_tryon(batch)
            if self.exu is not None:
                self.exu.launch()       # per bucket: wait for its in-graph signal -> all-reduce -> Adam, on exu's own stream
            gp.launch_warp_backward()
            if self.exw is not None:
                self.exw.launch()
            else:
                with gp.on_side():
                    self.redw.start()
                    self.optw.step(grad_scale=self.redw.finish())
            if self.exu is None:
                self.redu.start()
                self._u_started = True
            self._pending_u = True
            return gp.result_warp, gp.result_tryon
        if self.sync_buffers and not self.lazy_buffers:
            broadcast_buffers(self.warp)
        if self.exw is not None:
            self.exw.finish()
        rw = self._gw(batch)
        if self.exw is not None:
            self.exw.launch()
        else:
            self.redw.start()
        self.flush()
        if batch is not None:
            self._gu.load_batch({k: v for k, v in batch.items() if k != "cloth"})
        ru = self._gu()
        if self.exu is not None:
            self.exu.launch()
        else:
            self.redu.start()
            self._u_started = True
        self._pending_u = True
        if self.exw is None:
            self.optw.step(grad_scale=self.redw.finish())
        return rw, ru

    def synchronize(self):
        """Everything in flight lands; with lazy buffer broadcast this is also where rank 0's BatchNorm running statistics
        reach the other ranks (call it before validation / a checkpoint / reading the buffers).

        COLLECTIVE with several ranks (`lazy_buffers`): every rank must call it, in the same place - called on a subset of
        the ranks it deadlocks inside the broadcast.  After only flush() / torch.cuda.synchronize() the warp model's
        running_mean / running_var are still this rank's own; a rank-0-only checkpoint or validation path must go through
        synchronize() on ALL ranks first (Trainer.fit_chained does).  The per-step broadcast DDP pays is therefore not in
        bench.py's timed c4 region: the line says so in `config.lazy_buffers`."""
        self.flush()
        if self.exw is not None:
            self.exw.finish()
        if self._gp is not None:
            self._gp.join()
        if self.sync_buffers and self.lazy_buffers:
            broadcast_buffers(self.warp)
        torch.cuda.synchronize()


# ---- data: the dataset resident in HBM ----------------------------------------------------------------------------------
class MultiOptimizerStep:
    """One training batch of a model with SEVERAL optimizers, the way pytorch-lightning 0.9 drives it (the reference's
    SamsModel returns three from configure_optimizers, models/sams_model.py:130-145): for each optimizer in turn —
    only that optimizer's parameters require grad (Lightning toggles this so that a step never leaves gradients in the
    other networks), `training_step(batch, batch_idx, optimizer_idx)`, backward, gradient mean over the ranks,
    optimizer step, zero_grad.  Runs eagerly: the generator pass is five dependent forward passes whose launch count
    dwarfs a hipGraph's benefit only at toy sizes."""

    def __init__(self, model, optimizers, networks=None, accumulate=1, sync_buffers=True, bucketed=None,
                 bucket_bytes=64 << 20):
        self.model, self.optimizers = model, list(optimizers)
        # Lightning's DDP wrapper re-broadcasts rank 0's buffers before EVERY forward (broadcast_buffers=True), i.e. once
        # per optimizer_idx: the per-process "syncbatch" running statistics of every SPADE stay rank 0's on all ranks
        self.sync_buffers = bool(sync_buffers) and _collective() and flatten_float_buffers(model) is not None
        self.networks = list(networks) if networks is not None else model.optimizer_networks()
        if len(self.networks) != len(self.optimizers):
            raise ValueError("one network per optimizer")
        self.accumulate = max(1, accumulate)
        self._micro = 0
        self._all = [p for p in model.parameters()]
        self._frozen = [p for p in self._all if not p.requires_grad]  # e.g. the VGG of the perceptual loss: never toggled on
        self._own = [[p for p in net.parameters()] for net in self.networks]
        flats = [o.flat_grads for o in self.optimizers]  # builds the slabs now: backward then accumulates straight into them
        self.reducers = [GradientAllReducer(f) for f in flats] if _collective() else None
        # per optimizer: bucketed exchange overlapped with that optimizer's own backward pass (the update must have landed
        # before the next optimizer's forward, which reads the freshly updated network - Lightning's order)
        self.exchanges = None
        if self.accumulate == 1:
            ex = [_make_exchange(o, bucketed, bucket_bytes) for o in self.optimizers]
            self.exchanges = ex if all(e is not None for e in ex) else None
        self.stepped = False
        self._passes = getattr(model, "n_frames_now", None)

    def _only(self, idx):
        for p in self._all:
            p.requires_grad_(False)
        for p in self._own[idx]:
            p.requires_grad_(True)

    def restore_requires_grad(self):
        frozen = {id(p) for p in self._frozen}
        for p in self._all:
            p.requires_grad_(id(p) not in frozen)

    def _all_exchanges(self):
        return tuple(self.exchanges or ())

    @_abort_exchanges_on_error
    def __call__(self, batch, batch_idx=0):
        results = []
        passes = getattr(self.model, "n_frames_now", None)
        if passes != self._passes:   # progressive training changed the number of generator passes: new report counts
            self._passes = passes
            for ex in self._all_exchanges():
                if ex is not None:
                    ex.recalibrate()
        self._micro += 1
        update = self._micro % self.accumulate == 0
        for idx, opt in enumerate(self.optimizers):
            self._only(idx)
            if self.sync_buffers:
                broadcast_buffers(self.model)
            ex = self.exchanges[idx] if self.exchanges is not None else None
            if ex is not None:
                ex.begin()
            result = self.model.training_step(batch, batch_idx, idx)
            result.minimize.sum().backward()
            if ex is not None:
                ex.end()
                ex.launch()
                ex.finish()
                opt.zero_grad()
            elif update:
                scale = self.reducers[idx].all_reduce() if self.reducers is not None else 1.0
                opt.step(grad_scale=scale / self.accumulate)
                opt.zero_grad()
            results.append(result)
        self.stepped = update
        return results

    def flush(self):
        """Nothing is left in flight by an eager step; what a checkpoint / validation must not see is the requires_grad
        pattern of the last optimizer, so that is put back."""
        self.restore_requires_grad()


class DeviceBatches:
    """A map-style dataset collated ONCE and kept resident in HBM (288 GB: the synthetic VVT-shaped set is 9.4 MB per
    sample); batches are gathered by index on the device, so the input side costs one small kernel per batch key instead
    of a per-sample CPU pipeline.  Rank r of w takes indices r::w of the (optionally shuffled) epoch order, like
    DistributedSampler (base_model.py:113-121)."""

    def __init__(self, dataset, batch_size, device, shuffle=True, seed=0, limit=None, keys=None):
        """keys: keep only these tensor entries resident (the ones the model reads); None keeps all."""
        from torch.utils.data.dataloader import default_collate

        n = len(dataset) if limit is None else min(len(dataset), limit)
        full = default_collate([dataset[i] for i in range(n)])
        self.tensors = {k: v.to(device) for k, v in full.items()
                        if isinstance(v, torch.Tensor) and (keys is None or k in keys)}
        self.other = {k: v for k, v in full.items() if not isinstance(v, torch.Tensor)}
        self.n, self.batch_size, self.device, self.shuffle, self.seed = n, batch_size, device, shuffle, seed
        self.rank = dist.get_rank() if _world() > 1 else 0
        self.world = _world()
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        per_rank = (self.n + self.world - 1) // self.world
        return (per_rank + self.batch_size - 1) // self.batch_size

    def _select_strings(self, v, idx):
        if isinstance(v, (list, tuple)) and v and isinstance(v[0], (list, tuple)):
            return [self._select_strings(x, idx) for x in v]
        if isinstance(v, (list, tuple)) and len(v) == self.n:
            return [v[i] for i in idx]
        return v

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        order = torch.randperm(self.n, generator=g) if self.shuffle else torch.arange(self.n)
        if self.world > 1:  # pad to a multiple of the world size like DistributedSampler, then stride
            pad = (-self.n) % self.world
            order = torch.cat([order, order[:pad]])[self.rank::self.world]
        for i in range(0, len(order), self.batch_size):
            idx = order[i:i + self.batch_size]
            didx = idx.to(self.device)
            batch = {k: v.index_select(0, didx) for k, v in self.tensors.items()}
            batch.update({k: self._select_strings(v, idx.tolist()) for k, v in self.other.items()})
            yield batch


class Trainer:
    def __init__(self, gpus=None, distributed_backend="ddp", precision=32, default_root_dir="experiments",
                 accumulate_grad_batches=1, max_epochs=10, val_check_interval=1.0, limit_train_batches=1.0,
                 limit_val_batches=1.0, fast_dev_run=False, save_count=10000, resume_from_checkpoint=None,
                 broadcast_bn_buffers=True, graph=True, overlap=True, device_dataset="auto", **_):
        """graph: replay forward+backward as a hipGraph (static shapes; a ragged last batch runs eagerly).
        overlap: leave the gradient exchange + Adam of step i in flight until step i+1 is about to launch.
        device_dataset: keep the collated dataset resident in HBM ("auto": when the dataset asks for it through a
        `device_resident = True` attribute, as the synthetic one does)."""
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
        self.graph, self.overlap, self.device_dataset = graph, overlap, device_dataset
        self.global_step = 0
        self.current_epoch = 0
        self.model = self.optimizer = self.scheduler = None
        self.optimizers, self.schedulers = [], []
        self._interrupted = False
        self.epoch_seconds = []

    # ---- checkpoints (Lightning-style dict so reference checkpoints' state_dict drops in) --------------
    def save_checkpoint(self, path, epoch_finished=False):
        """`epoch` is the epoch a resumed run starts with: the current one for a mid-epoch checkpoint, the next one when
        the epoch has finished (Lightning stores current_epoch + 1)."""
        if self.rank != 0:
            return
        torch.cuda.synchronize()
        os.makedirs(osp.dirname(path) or ".", exist_ok=True)
        hp = vars(self.model.hparams)
        torch.save({
            "state_dict": {k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()},
            "hparams": hp, "hyper_parameters": hp,
            "optimizer_states": [o.state_dict() for o in self.optimizers],
            "lr_schedulers": [sc.state_dict() for sc in self.schedulers],
            "global_step": self.global_step, "epoch": self.current_epoch + (1 if epoch_finished else 0),
            "vgg_pretrained": _vgg_flag(self.model),
        }, path)

    def _maybe_resume(self):
        """Weights, optimizer moments, LR schedule position and counters all come back (train.py:39-54: the reference hands
        the same file to load_from_checkpoint and to resume_from_checkpoint)."""
        if not self.resume:
            return
        ckpt = torch.load(self.resume, map_location="cpu", weights_only=False)
        self.model.load_state_dict(ckpt["state_dict"], strict=True)
        vgg = getattr(_vgg_criterion(self.model), "vgg", None)
        if vgg is not None and any(k.startswith(("criterionVGG.", "criterion_VGG.")) for k in ckpt["state_dict"]):
            # the VGG weights now are the checkpoint's: pretrained unless the checkpoint says it was trained on random ones
            # (reference checkpoints carry no flag and always hold ImageNet weights, models/networks/vgg.py:9)
            vgg.pretrained_loaded = bool(ckpt.get("vgg_pretrained", True))
        self.global_step = int(ckpt.get("global_step", 0))
        self.current_epoch = int(ckpt.get("epoch", 0))
        states, scheds = ckpt.get("optimizer_states") or [], ckpt.get("lr_schedulers") or []
        for k, (optimizer, scheduler) in enumerate(zip(self.optimizers, self.schedulers)):
            if k < len(states):
                optimizer.load_state_dict(states[k])
            if k < len(scheds):
                scheduler.load_state_dict(dict(scheds[k]))
            else:  # older checkpoint: put the schedule at the epoch we resume with
                scheduler.last_epoch = self.current_epoch
            for group, lam, base in zip(optimizer.param_groups, scheduler.lr_lambdas, scheduler.base_lrs):
                group["lr"] = base * lam(scheduler.last_epoch)
            scheduler._last_lr = [g["lr"] for g in optimizer.param_groups]

    def _limit(self, n, lim):
        lim = str2num(str(lim))
        return min(n, lim) if isinstance(lim, int) else max(1, int(n * lim))

    def _loaders(self, model):
        want = self.device_dataset
        ds = getattr(model, "train_dataset", None)
        if want == "auto":
            want = bool(getattr(ds, "device_resident", False))
        if not want:
            return model.train_dataloader(), model.val_dataloader()
        hp = model.hparams
        shuffle = not getattr(hp, "no_shuffle", False)
        keys = model.batch_keys() if hasattr(model, "batch_keys") else None
        return (DeviceBatches(model.train_dataset, hp.batch_size, self.device, shuffle=shuffle, keys=keys),
                DeviceBatches(model.val_dataset, hp.batch_size, self.device, shuffle=False, keys=keys))

    def _on_sigint(self, *_):
        # only set a flag: the checkpoint is written at the next step boundary, after the streams have drained
        self._interrupted = True

    # ---- fit ------------------------------------------------------------------------------------------
    def fit(self, model):
        self.model = model.to(self.device)
        model.trainer = self
        model.prepare_data()
        model.setup("fit")
        self.optimizers, self.schedulers = (list(x) for x in model.configure_optimizers())
        self.optimizer, self.scheduler = self.optimizers[0], self.schedulers[0]
        multi = len(self.optimizers) > 1
        self._maybe_resume()
        if hasattr(model, "require_pretrained_vgg"):
            model.require_pretrained_vgg()  # after the resume: a checkpoint brings its own criterionVGG.* weights
        for optimizer in self.optimizers:
            broadcast_parameters(model, optimizer=optimizer)
        train_loader, val_loader = self._loaders(model)
        ckpt_dir = osp.join(self.root, "checkpoints")
        previous = signal.signal(signal.SIGINT, self._on_sigint)
        n_train = self._limit(len(train_loader), self.limit_train_batches)
        vci = str2num(str(self.val_check_interval))
        val_every = vci if isinstance(vci, int) and vci > 0 else max(1, int(n_train * (vci or 1)))
        step = None
        try:
            for epoch in range(self.current_epoch, self.max_epochs):
                self.current_epoch = model.current_epoch = epoch
                for obj in (train_loader, getattr(train_loader, "sampler", None)):
                    if hasattr(obj, "set_epoch"):
                        obj.set_epoch(epoch)
                model.train()
                t_epoch = time.perf_counter()
                for i, batch in enumerate(train_loader):
                    if i >= n_train or (self.fast_dev_run and i >= 1):
                        break
                    batch = _to_device(batch, self.device)
                    model.global_step = self.global_step
                    if step is None and multi:
                        step = MultiOptimizerStep(model, self.optimizers, accumulate=self.accumulate,
                                                  sync_buffers=self.broadcast_bn_buffers)
                    elif step is None:
                        step = TrainStep(model, self.optimizer, batch, graph=self.graph, overlap=self.overlap,
                                         accumulate=self.accumulate, sync_buffers=self.broadcast_bn_buffers)
                    self.last_result = step(batch, i) if multi else step(batch)
                    if step.stepped:
                        self.global_step += 1
                        if self.global_step % self.save_count == 0:
                            step.flush()
                            self.save_checkpoint(osp.join(ckpt_dir, f"step_{self.global_step:09d}.ckpt"))
                    if self._interrupted:
                        step.flush()
                        self.save_checkpoint(osp.join(ckpt_dir, "interrupted_by_Ctrl-C.ckpt"))
                        raise KeyboardInterrupt
                    if (i + 1) % val_every == 0:
                        step.flush()
                        self._validate(model, val_loader)
                        model.train()
                if step is not None:
                    step.flush()
                torch.cuda.synchronize()   # once per epoch: wall time of the epoch for throughput reports
                self.epoch_seconds.append(time.perf_counter() - t_epoch)
                for scheduler in self.schedulers:
                    scheduler.step()
        except KeyboardInterrupt:
            raise SystemExit(130)
        except Exception as e:  # mirror train.py:63-66: checkpoint, then re-raise
            try:  # land the exchange + Adam still in flight (overlap=True) and restore the requires_grad pattern, so the
                if step is not None:  # checkpoint holds the parameters / moments of the step global_step counts
                    step.flush()
            except Exception:  # noqa: BLE001 - the original error is the one to report
                logger.exception("flush before the failure checkpoint failed; the checkpoint may lag one step")
            torch.cuda.synchronize()
            bad = exchange_statuses(step) if step is not None else {}
            if bad:   # a bucket wait was aborted / timed out: Adam ran on incomplete gradients - say so in the file name
                logger.error("gradient-bucket waits ended abnormally %s: the weights / moments of the last step are NOT a valid "
                             "training state; writing the checkpoint as *.incomplete_exchange.ckpt", bad)
            self.save_checkpoint(osp.join(ckpt_dir, f"interrupted_by_{type(e).__name__}" +
                                          (".incomplete_exchange" if bad else "") + ".ckpt"))
            raise
        finally:
            signal.signal(signal.SIGINT, previous)
        self.save_checkpoint(osp.join(ckpt_dir, "final.ckpt"), epoch_finished=True)

    def build_chained_step(self, warp, unet, sample_batch, schedule="auto", log=None):
        """The chained warp -> try-on engine on this trainer's device: optimizers built, rank 0's weights broadcast.
        This is what bench.py times."""
        warp, unet = warp.to(self.device).train(), unet.to(self.device).train()
        (optw,), _ = warp.configure_optimizers()
        (optu,), _ = unet.configure_optimizers()
        broadcast_parameters(warp, optimizer=optw)
        broadcast_parameters(unet, optimizer=optu)
        schedule = schedule if self.graph or schedule == "eager" else "eager"
        return ChainedTrainStep(warp, optw, unet, optu, sample_batch, schedule=schedule,
                                sync_buffers=self.broadcast_bn_buffers, log=log)

    def fit_chained(self, warp, unet, batches, steps):
        """Train both stages in lockstep on `batches` (an iterable of device batches of one fixed shape) for `steps`
        chained steps: the C4 workload through the product API.  Returns the engine (losses: engine results)."""
        engine = None
        done = 0
        while done < steps:
            for batch in batches:
                if engine is None:
                    engine = self.build_chained_step(warp, unet, batch)
                engine(batch)
                done += 1
                self.global_step += 1
                if done >= steps:
                    break
        engine.synchronize()
        return engine

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


def _vgg_criterion(model):
    """The perceptual-loss module: `criterionVGG` in UnetMaskModel, `criterion_VGG` in SamsModel (the reference's names)."""
    crit = getattr(model, "criterionVGG", None)
    return crit if crit is not None else getattr(model, "criterion_VGG", None)


def _vgg_flag(model):
    crit = _vgg_criterion(model)
    return bool(getattr(getattr(crit, "vgg", None), "pretrained_loaded", False)) if crit is not None else None

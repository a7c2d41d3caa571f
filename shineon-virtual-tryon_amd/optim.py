"""Adam on flat fp32 slabs, one fused HIP launch per step (reference: torch.optim.Adam created at
models/base_model.py:165-168, defaults betas=(0.9, 0.999), eps=1e-8, no weight decay).

All trainable parameters are re-homed as views into one contiguous slab (their logical shapes, strides
and state_dict entries are unchanged), and so are their gradients: autograd accumulates straight into
the flat gradient slab, which is also the buffer the data-parallel all-reduce works on.
"""
import torch

from . import ops


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, adjacent=()):
        """adjacent: tuples of parameters that must follow each other (in that order) in the flat slabs."""
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("HipAdam got no trainable parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._flat = None
        self._slots = []
        self._steps = 0
        self._adjacent = [tuple(g) for g in adjacent]

    # ---- flat slabs ---------------------------------------------------------------------------------
    def _params(self):
        return [p for g in self.param_groups for p in g["params"]]

    def _build(self):
        ps = _slab_order(self._params(), self._adjacent)
        dev = ps[0].device
        if dev.type != "cuda":
            raise RuntimeError("HipAdam needs parameters on an MI355X (model.cuda() first); no CPU fallback")
        sizes = [(p.numel() + 3) // 4 * 4 for p in ps]  # keep every view 16-byte aligned
        total = sum(sizes)
        flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        flat_g = torch.empty(total, dtype=torch.float32, device=dev)
        ops.fill_(flat_p, 0.0)
        ops.fill_(flat_g, 0.0)
        off = 0
        self._slots = []  # (parameter, slab offset, data_ptr of its view, data_ptr of its gradient view)
        for p, sz in zip(ps, sizes):
            n = p.numel()
            # a view with the parameter's own (possibly channels_last) strides over the slab
            view = torch.as_strided(flat_p, p.shape, p.stride(), off)
            dense = _is_dense(p)
            if not dense:
                view = flat_p[off:off + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            gview = torch.as_strided(flat_g, p.shape, view.stride(), off)
            if p.grad is not None:  # built lazily after a first backward: keep what autograd already produced
                gview.copy_(p.grad)
            p.grad = gview
            p._so_grad_direct = True  # ops.py may accumulate wgrad / bias-grad kernels straight into this view
            self._slots.append((p, off, view.data_ptr(), gview.data_ptr()))
            off += sz
        self._flat = (flat_p, flat_g, ops.fill_(torch.empty_like(flat_p), 0.0), ops.fill_(torch.empty_like(flat_p), 0.0))

    def layout_signature(self):
        """(numel, ...) of the parameters in slab order: stored next to the flat moments in a checkpoint and verified on
        load, so moments are never copied into a slab that is laid out differently."""
        if self._flat is None:
            self._build()
        return [int(p.numel()) for p, *_ in self._slots]

    def check_slabs(self):
        """The flat-slab contract: every parameter's .data and .grad still ARE the views planted by _build().  Anything
        that re-homes them (model.to(), .float(), model.zero_grad(set_to_none=True), an optimizer that assigns
        p.grad = ...) would leave step() updating memory autograd no longer writes to - raise instead of training on
        stale zeros."""
        for p, _, dptr, gptr in self._slots:
            if p.data_ptr() != dptr:
                raise RuntimeError("HipAdam: a parameter was moved out of the optimizer's flat slab after the optimizer was "
                                   "built (model.to() / .float() / .cuda() later than configure_optimizers?). Rebuild the optimizer.")
            if p.grad is None or p.grad.data_ptr() != gptr:
                raise RuntimeError("HipAdam: a parameter's .grad is no longer the optimizer's flat gradient view (was "
                                   "model.zero_grad() or zero_grad(set_to_none=True) called? use optimizer.zero_grad()).")

    @property
    def flat_params(self):
        if self._flat is None:
            self._build()
        return self._flat[0]

    @property
    def flat_grads(self):
        if self._flat is None:
            self._build()
        return self._flat[1]

    def zero_grad(self, set_to_none=False):
        """Gradients stay views of the flat slab; one fill kernel clears them."""
        ops.fill_(self.flat_grads, 0.0)

    def slot_table(self):
        """[(parameter, slab offset, padded size in floats)] in slab order (trainer.BucketedExchange cuts buckets from it)."""
        if self._flat is None:
            self._build()
        offs = [off for _, off, _, _ in self._slots] + [self._flat[0].numel()]
        return [(p, off, offs[i + 1] - off) for i, (p, off, _, _) in enumerate(self._slots)]

    @torch.no_grad()
    def begin_step(self):
        """Advance the step counter once; the caller then applies the update piecewise with step_range()."""
        if self._flat is None:
            self._build()
        if not torch.cuda.is_current_stream_capturing():
            self.check_slabs()
        self._steps += 1

    @torch.no_grad()
    def step_range(self, lo, hi, grad_scale=1.0):
        """Adam on the slab range [lo, hi) with the step number set by begin_step(): element-wise, so any partition of the
        slab gives bit-identical parameters to one step() over the whole of it."""
        flat_p, flat_g, m, v = self._flat
        group = self.param_groups[0]
        b1, b2 = group["betas"]
        ops.adam_step(flat_p[lo:hi], flat_g[lo:hi], m[lo:hi], v[lo:hi], float(group["lr"]), b1, b2, group["eps"], self._steps,
                      grad_scale)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        if self._flat is None:
            self._build()
        if not torch.cuda.is_current_stream_capturing():
            self.check_slabs()
        flat_p, flat_g, m, v = self._flat
        group = self.param_groups[0]
        self._steps += 1
        b1, b2 = group["betas"]
        ops.adam_step(flat_p, flat_g, m, v, float(group["lr"]), b1, b2, group["eps"], self._steps, grad_scale)
        return loss

    # ---- checkpointing ------------------------------------------------------------------------------
    _GROUP_KEYS = ("lr", "betas", "eps", "initial_lr")

    def state_dict(self):
        sd = {"param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
              "steps": self._steps}
        if self._flat is not None:
            sd["exp_avg"], sd["exp_avg_sq"] = self._flat[2].cpu(), self._flat[3].cpu()
            sd["layout"] = self.layout_signature()
        return sd

    def load_state_dict(self, sd):
        """Accepts this class's own format (flat moments + layout signature) and torch.optim.Adam's
        ({"state": {index: {step, exp_avg, exp_avg_sq}}, "param_groups": [{"params": [indices], ...}]}, what a
        reference / Lightning checkpoint carries).  Hyper-parameters are taken from the checkpoint; the parameter
        lists themselves never are."""
        for g, s_ in zip(self.param_groups, sd.get("param_groups", [])):
            for k in self._GROUP_KEYS:
                if k in s_:
                    g[k] = tuple(s_[k]) if k == "betas" else s_[k]
        if "state" in sd:
            return self._load_torch_adam_state(sd)
        self._steps = int(sd.get("steps", 0))
        if "exp_avg" in sd:
            if self._flat is None:
                self._build()
            if "layout" in sd and list(sd["layout"]) != self.layout_signature():
                raise RuntimeError("HipAdam.load_state_dict: the checkpoint's slab layout (parameter sizes in slab order) does "
                                   "not match this model's; refusing to copy the Adam moments")
            if sd["exp_avg"].numel() != self._flat[2].numel():
                raise RuntimeError("HipAdam.load_state_dict: flat moment size mismatch")
            self._flat[2].copy_(sd["exp_avg"])
            self._flat[3].copy_(sd["exp_avg_sq"])

    def _load_torch_adam_state(self, sd):
        """Per-parameter torch.optim.Adam moments -> the flat slabs.  torch indexes the parameters in the order they were
        handed to the optimizer; the reference hands over self.parameters() (base_model.py:165-168), of which the frozen
        VGG tail has no state - the trainable ones come first and in the same order as ours."""
        if self._flat is None:
            self._build()
        params = self._params()
        slot_of = {id(p): (off, p) for p, off, _, _ in self._slots}
        state = {int(k): v for k, v in sd["state"].items()}
        steps = set()
        for idx, st in state.items():
            if idx >= len(params):
                raise RuntimeError(f"optimizer state for parameter #{idx}, but only {len(params)} parameters are optimised")
            off, p = slot_of[id(params[idx])]
            for key, flat in (("exp_avg", self._flat[2]), ("exp_avg_sq", self._flat[3])):
                src = st[key]
                if tuple(src.shape) != tuple(p.shape):
                    raise RuntimeError(f"optimizer state #{idx} has shape {tuple(src.shape)}, parameter has {tuple(p.shape)}")
                # the moments live in the slab with the parameter's own (e.g. OHWI) strides
                torch.as_strided(flat, p.shape, p.stride(), off).copy_(src.to(flat.device, torch.float32))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise RuntimeError(f"per-parameter Adam step counters differ ({sorted(steps)}); cannot map onto one fused step")
        missing = len(params) - len(state)
        if state and missing:
            raise RuntimeError(f"{missing} optimised parameters have no state in the checkpoint")
        self._steps = steps.pop() if steps else 0


def _slab_order(ps, adjacent=()):
    """Slab layout order: the parameters of every group in `adjacent` are placed next to each other in the group's
    order, at the position of the group's first member (SelfAttention lays its query/key/value weights - and biases -
    out as one [2d+C, C] matrix so that one GEMM serves the three projections); everything else keeps the optimizer's
    order.  Groups with a member that is not being optimized are ignored."""
    known = {id(p) for p in ps}
    group_of = {}
    for g in adjacent:
        if all(id(p) in known for p in g):
            for p in g:
                group_of[id(p)] = g
    out, placed = [], set()
    for p in ps:
        if id(p) in placed:
            continue
        g = group_of.get(id(p))
        for q in (g if g is not None else (p,)):
            out.append(q)
            placed.add(id(q))
    return out


def _is_dense(p):
    """True if the parameter's strides describe a dense permutation of numel() elements."""
    expected = 1
    for size, stride in sorted(zip(p.shape, p.stride()), key=lambda t: t[1]):
        if size == 1:
            continue
        if stride != expected:
            return False
        expected *= size
    return True

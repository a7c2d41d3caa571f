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
            p.grad = gview
            p._so_grad_direct = True  # ops.py may accumulate wgrad / bias-grad kernels straight into this view
            off += sz
        self._flat = (flat_p, flat_g, ops.fill_(torch.empty_like(flat_p), 0.0), ops.fill_(torch.empty_like(flat_p), 0.0))

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

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        if self._flat is None:
            self._build()
        flat_p, flat_g, m, v = self._flat
        group = self.param_groups[0]
        self._steps += 1
        b1, b2 = group["betas"]
        ops.adam_step(flat_p, flat_g, m, v, float(group["lr"]), b1, b2, group["eps"], self._steps, grad_scale)
        return loss

    # ---- checkpointing ------------------------------------------------------------------------------
    def state_dict(self):
        sd = {"param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
              "steps": self._steps}
        if self._flat is not None:
            sd["exp_avg"], sd["exp_avg_sq"] = self._flat[2].cpu(), self._flat[3].cpu()
        return sd

    def load_state_dict(self, sd):
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
        self._steps = sd.get("steps", 0)
        if "exp_avg" in sd:
            if self._flat is None:
                self._build()
            self._flat[2].copy_(sd["exp_avg"])
            self._flat[3].copy_(sd["exp_avg_sq"])


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

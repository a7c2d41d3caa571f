"""Autograd operators of the try-on hot path, each one a thin host wrapper around the C ABI
(include/shineon_hip.h).  torch is used for device memory (torch.empty), the current stream and the
autograd tape only; every FLOP and every byte moved on the hot path goes through libshineon_hip.so.

Layout contract: a 4-D activation is a torch tensor of logical shape (N, C, H, W) whose strides are
NHWC-with-pitch: (H*W*ld, 1, W*ld, ld), ld >= C.  That is exactly torch's channels_last when ld == C,
and a channel slice of a wider NHWC buffer when ld > C, so the reference's NCHW-shaped module
interfaces are kept while the kernels see "pixel rows x channel columns".
"""
import math
import os
import weakref

import torch

from ._lib import check, lib

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_GELU, ACT_SWISH, ACT_SINE, ACT_TANH, ACT_SIGMOID = range(8)
ACT_CODES = {
    None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "leaky": ACT_LEAKY, "gelu": ACT_GELU,
    "swish": ACT_SWISH, "sine": ACT_SINE, "tanh": ACT_TANH, "sigmoid": ACT_SIGMOID,
}

_WS = {}
_WS_RETIRED = []  # outgrown slabs, kept alive for graphs / streams that captured their address
_WS_BYTES = 256 << 20


def _require_cuda(t):
    if not t.is_cuda:
        raise RuntimeError(
            "shineon_amd ops run on an MI355X only (got a CPU tensor); there is no CPU fallback. "
            "Use oracle/ for a CPU check."
        )


def _stream():
    return torch.cuda.current_stream().cuda_stream


_LANE_BASE = [0]

# trainer.BucketedExchange installs a callback here while a backward pass is issued (eagerly or under stream capture): every
# operator that has just launched the kernels producing a parameter's gradient reports that parameter, so the exchange of a
# gradient bucket can start as soon as the backward pass has produced it (DDP's "bucket ready" hooks).
_GRAD_READY = [None]


def grad_ready(*params):
    cb = _GRAD_READY[0]
    if cb is not None:
        for q in params:
            if q is not None:
                cb(q)


class workspace_lane:
    """Context manager: kernels issued inside use their own scratch slabs (lane offset), so that two branches of work
    running concurrently on different streams never share split-K slabs or reduction partials."""

    def __init__(self, base):
        self.base = base

    def __enter__(self):
        self.prev = _LANE_BASE[0]
        _LANE_BASE[0] = self.base

    def __exit__(self, *exc):
        _LANE_BASE[0] = self.prev


def workspace(device, min_bytes=0, lane=0):
    """Per-device scratch slab (split-K slabs, reduction partials).  `lane` selects an independent slab for work
    issued on the side stream (weight gradients running concurrently with input gradients)."""
    key = (device.type, device.index, lane + _LANE_BASE[0])
    ws = _WS.get(key)
    need = max(_WS_BYTES, int(min_bytes))
    if ws is None or ws.numel() * 4 < need:
        if ws is not None:
            # A slab that has been handed out is never freed: captured hipGraphs hold its raw pointer and the side
            # stream may still be using it.  Growing means a NEW, larger slab for later callers; the old one stays alive.
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f"workspace lane {key[2]} would have to grow to {need} bytes during a stream capture; "
                                   "run one eager step first so every scratch size is known before capturing")
            _WS_RETIRED.append(ws)
        ws = torch.empty(need // 4, dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


_SIDE = {}
# Input gradients of trainable convolutions read the OHWI weights in place (the engine stages them untransposed; a transposed
# copy per step measured 0.05 ms/step slower and was removed).  The frozen VGG chain uses cached transposed weights.
# Split-bf16 convolutions for the FROZEN VGG19 chain (csrc/sb16.hip): fp32 = hi + mid bf16 planes, three bf16 MFMAs per
# product instead of one fp32 MFMA stream at 1/16 of the rate.  Not bit-equal to the fp32 path (error ~3x the fp32 MFMA
# chain's own round-off), so it is OFF by default and reported as its own bench line (bench.py --vgg-split-bf16).
VGG_SPLIT_BF16 = os.environ.get("SHINEON_VGG_SPLIT_BF16", "0") == "1"
# Winograd F(2x2, 3x3) for the frozen VGG19 chain (csrc/wino.hip): fp32 arithmetic, 2.25x fewer MFMA FLOPs.
#   "fused"    one launch: input transform at fragment-build time, 16 GEMMs, output transform through LDS
#   "nonfused" transform / batched GEMM / transform (pays 4x the activation size in transformed operands)
#   "auto"     per layer from the measured crossover (tools/wino_bench.py)
#   "0"        direct implicit GEMM everywhere
WINOGRAD = os.environ.get("SHINEON_WINOGRAD", "auto")
WINOGRAD_F44 = os.environ.get("SHINEON_WINOGRAD_F44", "1") != "0"   # F(4x4,3x3) for the deep layers (csrc/wino.hip)
# Trainable weights must be re-transformed every call (100 bytes of traffic per (ko, c) filter): on the 768-pixel layers
# (GMM 3x3 at 16x12, U-Net u4 / u5) that costs what the transform saves - kernel trace r03_h: 17 weight transforms per step =
# 0.29 ms.  Winograd for trainable convolutions therefore starts at 3072 pixels (32x24 at bs = 4).
WINOGRAD_TRAINABLE_MIN_PIXELS = int(os.environ.get("SHINEON_WINOGRAD_TRAINABLE_MIN_PIXELS", "3072"))



class _SideStream:
    """Fork/join helper: `with side(device) as st:` runs the body on a second HIP stream that first waits for
    the current stream and is joined back by `join()`.  Works eagerly and inside a stream capture (the fork and
    join become graph edges).  Tensors used in the body must stay alive until join() has been called."""

    def __init__(self, device):
        self.cur = torch.cuda.current_stream(device)
        key = (device.index, self.cur.cuda_stream)
        if key not in _SIDE:
            _SIDE[key] = torch.cuda.Stream(device)
        self.side = _SIDE[key]
        self.ctx = None

    def __enter__(self):
        self.side.wait_stream(self.cur)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self.side

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        return False

    def join(self):
        self.cur.wait_stream(self.side)


def nhwc_empty(n, h, w, c, device, ld=None):
    """(N, C, H, W)-shaped view over a fresh [N][H][W][ld] buffer."""
    ld = c if ld is None else ld
    buf = torch.empty((n, h, w, ld), dtype=torch.float32, device=device)
    t = buf.permute(0, 3, 1, 2)
    return t if ld == c else t[:, :c]


def _is_rows(t):
    if t.dim() != 4 or t.dtype != torch.float32:
        return False
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    if c > 1 and sc != 1:
        return False
    ld = sw if w > 1 else (sh if h > 1 else (sn if n > 1 else c))
    if ld < c:
        return False
    if w > 1 and sw != ld:
        return False
    if h > 1 and sh != w * ld:
        return False
    if n > 1 and sn != h * w * ld:
        return False
    return True


def _ld(t):
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    if w > 1:
        return sw
    if h > 1:
        return sh
    if n > 1:
        return sn
    return c


def to_rows(t, cpad=None):
    """Return a tensor in the NHWC-with-pitch layout (converting with our own kernels if needed).
    cpad: produce a fresh buffer whose channel count is zero-padded up to cpad."""
    _require_cuda(t)
    if t.dtype != torch.float32:
        raise TypeError("fp32 only")
    n, c, h, w = t.shape
    L = lib()
    if cpad is None or cpad == c:
        if _is_rows(t):
            return t
        cd = c
    else:
        cd = cpad
    out = nhwc_empty(n, h, w, cd, t.device)
    if _is_rows(t):
        check(L.so_copy2d(t.data_ptr(), _ld(t), c, out.data_ptr(), cd, cd, n * h * w, 0, _stream()), "copy2d")
    else:
        src = t if t.is_contiguous() else t.contiguous()
        check(L.so_nchw_to_nhwc(src.data_ptr(), out.data_ptr(), cd, n, c, cd, h * w, _stream()), "nchw_to_nhwc")
    return out


def to_nchw(t):
    """Planar NCHW copy of an NHWC-pitch tensor (boundary conversion for PNG writers / visualisation)."""
    _require_cuda(t)
    if t.is_contiguous():
        return t
    t = to_rows(t)
    n, c, h, w = t.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=t.device)
    check(lib().so_nhwc_to_nchw(t.data_ptr(), _ld(t), out.data_ptr(), n, c, h * w, _stream()), "nhwc_to_nchw")
    return out


def _aligned(t, ld):
    return (t.data_ptr() % 16 == 0) and (ld % 4 == 0)


def _dense_rows(t):
    """rows layout AND 16-byte aligned / pitch % 4 == 0 (what the MFMA loaders need)."""
    t = to_rows(t)
    if not _aligned(t, _ld(t)) or t.shape[1] % 4:
        c = t.shape[1]
        t = to_rows(t, cpad=(c + 3) // 4 * 4) if c % 4 else _copy_rows(t)
    return t


def _copy_rows(t):
    n, c, h, w = t.shape
    out = nhwc_empty(n, h, w, c, t.device)
    check(lib().so_copy2d(t.data_ptr(), _ld(t), c, out.data_ptr(), c, c, n * h * w, 0, _stream()), "copy2d")
    return out


def _ohwi(weight, cpad=None):
    """Weight (O, I, R, S) -> dense OHWI memory [O][R][S][Ipad] as an (O, R, S, Ipad) tensor."""
    o, i, r, s = weight.shape
    w = weight.detach()
    cp = i if cpad is None else cpad
    L = lib()
    ohwi_view = w.permute(0, 2, 3, 1)
    if ohwi_view.is_contiguous():
        if cp == i:
            return ohwi_view
        out = torch.empty((o, r, s, cp), dtype=torch.float32, device=w.device)
        check(L.so_copy2d(w.data_ptr(), i, i, out.data_ptr(), cp, cp, o * r * s, 0, _stream()), "copy2d")
        return out
    src = w if w.is_contiguous() else w.contiguous()
    out = torch.empty((o, r, s, cp), dtype=torch.float32, device=w.device)
    check(L.so_nchw_to_nhwc(src.data_ptr(), out.data_ptr(), cp, o, i, cp, r * s, _stream()), "nchw_to_nhwc")
    return out


_IHWO_CACHE = {}
_CAPTURE_PINS = {}          # data_ptr -> derived-weight copy read by a graph captured outside graphs.py: never freed
_PIN_SINK = [None]          # graphs.py: the list of the step being captured - its pins live exactly as long as the step
_CACHE_GENERATION = [0]     # bumped by invalidate_weight_caches(); graphs.py refuses to replay a step captured before the bump


def _serve_cached(*tensors):
    """Hand a cache hit to the caller.  Under a stream capture the copy is pinned: the graph keeps only its address, and
    invalidate_weight_caches() must not free memory a captured step still reads.  The pin belongs to the step being captured
    (capture_pins) or, for a bare torch.cuda.graph, to the process."""
    if torch.cuda.is_current_stream_capturing():
        sink = _PIN_SINK[0]
        for t in tensors:
            if sink is not None:
                sink.append(t)
            else:
                _CAPTURE_PINS.setdefault(t.data_ptr(), t)
    return tensors[0] if len(tensors) == 1 else tensors


class capture_pins:
    """with ops.capture_pins(step_pins): ...capture...  - cache copies served during the capture are appended to `step_pins`."""

    def __init__(self, sink):
        self.sink = sink

    def __enter__(self):
        self.prev, _PIN_SINK[0] = _PIN_SINK[0], self.sink
        return self.sink

    def __exit__(self, *exc):
        _PIN_SINK[0] = self.prev
        return False


def cache_generation():
    return _CACHE_GENERATION[0]


def _ihwo(w_ohwi, owner=None):
    """(O, R, S, C) dense OHWI weights -> (C, R, S, O) transposed copy for the KC x KC input-gradient GEMM.
    owner (for frozen weights): (weakref to the owning parameter, its _version at forward time); the copy is
    cached per parameter OBJECT and version, never per address (addresses are recycled between models)."""
    # Under a stream capture an entry is never CREATED (its transform would be recorded, not run - an eager caller could read it
    # before the first replay), but entries made by earlier eager launches are served and pinned (_serve_cached): the warm-up
    # passes in front of a capture fill the caches, so the frozen VGG's derived weights cost a replayed step nothing.
    capturing = torch.cuda.is_current_stream_capturing()
    if owner is not None:
        ref, version = owner
        p = ref()
        hit = _IHWO_CACHE.get(id(p)) if p is not None else None
        if hit is not None and hit[0]() is p and hit[1] == (version, p.data_ptr()) and hit[2].shape[0] == w_ohwi.shape[3]:
            return _serve_cached(hit[2])
    o, r, s, c = w_ohwi.shape
    wt = torch.empty((c, r, s, o), dtype=torch.float32, device=w_ohwi.device)
    check(lib().so_ohwi_to_ihwo(w_ohwi.data_ptr(), wt.data_ptr(), o, r * s, c, _stream()), "ohwi_to_ihwo")
    if owner is not None and owner[0]() is not None and not capturing:
        dead = [k for k, v in _IHWO_CACHE.items() if v[0]() is None]
        for k in dead:
            del _IHWO_CACHE[k]
        _IHWO_CACHE[id(owner[0]())] = (owner[0], (owner[1], owner[0]().data_ptr()), wt)
    return wt


def _colsum(t2d_ptr, ld, rows, c, device):
    L = lib()
    out = torch.empty(c, dtype=torch.float32, device=device)
    need = L.so_colsum_ws_floats(rows, c) * 4
    ws = workspace(device, need)
    check(L.so_colsum(t2d_ptr, ld, rows, c, out.data_ptr(), 0, ws.data_ptr(), _stream()), "colsum")
    return out


def _direct_grad_ok(p, ohwi):
    """True if `p` is a parameter whose .grad was planted by HipAdam as a dense view of the flat gradient
    slab (same memory order as the kernels produce), so kernels may accumulate into it directly."""
    if not getattr(p, "_so_grad_direct", False) or p.grad is None or not p.requires_grad:
        return False
    g = p.grad
    if g.dtype != torch.float32 or g.data_ptr() % 16:
        return False
    return g.permute(0, 2, 3, 1).is_contiguous() if ohwi else g.is_contiguous()


# ------------------------------------------------------------------------------------------------
# Conv2d
# ------------------------------------------------------------------------------------------------
def _pad_rows(w_ohwi, o_pad):
    """(O, R, S, C) -> (O_pad, R, S, C) with zero rows appended (output-channel padding of the weights)."""
    o, r, s, c = w_ohwi.shape
    if o_pad == o:
        return w_ohwi
    out = torch.empty((o_pad, r, s, c), dtype=torch.float32, device=w_ohwi.device)
    fill_(out, 0.0)
    check(lib().so_copy2d(w_ohwi.data_ptr(), c, c, out.data_ptr(), c, c, o * r * s, 0, _stream()), "copy2d")
    return out


class _Conv2dFn(torch.autograd.Function):
    """Conv2d on the MFMA implicit-GEMM engine.  Channel counts that are not multiples of 4 (n_frames_total > 1
    gives ngf = 134, 167, ...) are handled by zero padding: input channels of x / w are padded to cp, output
    channels are computed as op = ceil4(o) columns of which the last op - o are exactly zero."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act, zero_bias_grad, act_grad_external=False, act_param=0.0,
                bias_grad_hint=False, forward_only=False, bias_grad_external=False):
        L = lib()
        ctx.bias_hint = bool(bias_grad_hint)
        ctx.bias_external = bool(bias_grad_external)
        if act not in (ACT_NONE, ACT_RELU, ACT_LEAKY):
            # the backward pass evaluates act' from the OUTPUT, which only sign-preserving piecewise-linear maps allow
            raise ValueError("conv2d fuses ReLU / LeakyReLU only")
        o, i, r, s = weight.shape
        cp, op = (i + 3) // 4 * 4, (o + 3) // 4 * 4
        if cp == i:
            xr = _dense_rows(x)
        elif getattr(x, "_so_zero_padded", 0) == cp and _is_rows(x) and _ld(x) == cp and x.data_ptr() % 16 == 0:
            # the caller already holds this input as a channel slice of a zero-padded cp-wide buffer (label maps shared by
            # many SPADE convolutions): read that buffer instead of padding a private copy per convolution
            xr = torch.as_strided(x, (x.shape[0], cp, x.shape[2], x.shape[3]), x.stride(), x.storage_offset())
        else:
            xr = to_rows(x, cpad=cp)
        w = _ohwi(weight, cpad=cp)
        n, _, h, wd = xr.shape
        ho = (h + 2 * pad - r) // stride + 1
        wo = (wd + 2 * pad - s) // stride + 1
        y = nhwc_empty(n, ho, wo, op, x.device)
        ws = workspace(x.device)
        wino = _wino_mode(cp, op, n, h, wd) if (r == 3 and s == 3 and stride == 1 and pad == 1 and
                                                  n * h * wd >= WINOGRAD_TRAINABLE_MIN_PIXELS) else "direct"
        if wino != "direct":
            # 3x3 / s1 / p1 (U-Net up path, SPADE): Winograd F(2x2,3x3); trainable weights are transformed per call
            # forward only on a stand-alone parameter (inference: every update goes through torch and bumps its version): the
            # Winograd-domain weights are cached like the frozen VGG's; parameters planted in an optimizer's slab are updated
            # through raw pointers, so they - and every pass that needs gradients - are transformed per call
            owner = ((weakref.ref(weight), weight._version)
                     if forward_only and weight.untyped_storage().nbytes() == weight.numel() * 4 else None)
            u = _wino_weights(w, owner, False, fused=wino == "fused", ko_pad=op, f44=wino == "nonfused4")
            wino_conv3x3(xr.data_ptr(), _ld(xr), u, bias, None, y.data_ptr(), op, n, h, wd, cp, op, act, x.device,
                         fused=wino == "fused", act_param=act_param, f44=wino == "nonfused4")
        else:
            check(
                L.so_conv2d_fprop_padded(
                    xr.data_ptr(), _ld(xr), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                    y.data_ptr(), op, n, h, wd, cp, op, o, r, s, stride, pad, act, float(act_param),
                    ws.data_ptr(), ws.numel() * 4, _stream(),
                ),
                "conv2d_fprop",
            )
        ctx.save_for_backward(xr, w, y if act != ACT_NONE and not act_grad_external else None)
        # act_grad_external: the consumer (a BatchNorm told so) applies the activation's mask to the gradient it sends
        ctx.cfg = (stride, pad, ACT_NONE if act_grad_external else act, i, cp, op, bias is not None, tuple(weight.shape),
                   zero_bias_grad, float(act_param))
        # parameters whose .grad is a view of the optimizer's flat slab get their gradient accumulated in place
        ctx.direct = (weight if _direct_grad_ok(weight, ohwi=True) else None,
                      bias if bias is not None and _direct_grad_ok(bias, ohwi=False) else None)
        return y if op == o else y[:, :o]

    @staticmethod
    def backward(ctx, dy):
        L = lib()
        xr, w, y = ctx.saved_tensors
        stride, pad, act, i, cp, op, has_bias, wshape, zero_bias_grad, act_param = ctx.cfg
        # bias_grad_hint: the only consumer of the output (ops_sams._SpadeFn) reduces its gradient's columns while it
        # writes it and hands the sums over on the gradient tensor itself
        hint = getattr(dy, "_so_bias_grad", None) if ctx.bias_hint else None
        o, _, r, s = wshape
        n, _, h, wd = xr.shape
        dy = to_rows(dy, cpad=op) if op != o else _dense_rows(dy)  # [rows][op], pad columns zero
        dev = dy.device
        rows_out = dy.shape[0] * dy.shape[2] * dy.shape[3]
        if act != ACT_NONE:
            g = nhwc_empty(dy.shape[0], dy.shape[2], dy.shape[3], op, dev)
            check(L.so_act_bwd(y.data_ptr(), op, dy.data_ptr(), _ld(dy), g.data_ptr(), op, rows_out, op, act, act_param, _stream()), "act_bwd")
            dy = g
        dx = dw = db = None
        w_direct, b_direct = ctx.direct
        need_w = ctx.needs_input_grad[1]
        need_b = has_bias and ctx.needs_input_grad[2]
        if ctx.bias_external:
            need_b = False   # the BatchNorm behind this convolution's ReLU accumulates it (batch_norm_train(conv_bias=...))
        plain = cp == i and op == o  # no channel padding anywhere: kernels can write the slab layout directly
        keep = []  # temporaries of the side stream must outlive the join

        def weight_grads(lane):
            """dW (+ d bias) on the current stream with the scratch slab of `lane`."""
            dw_, db_ = None, None
            wsl = workspace(dev, lane=lane)
            if need_w:
                if w_direct is not None and plain:
                    # dW += ... straight into the flat gradient slab (split-K reduce / epilogue adds the old value)
                    check(
                        L.so_conv2d_wgrad_acc(dy.data_ptr(), _ld(dy), xr.data_ptr(), _ld(xr), w_direct.grad.data_ptr(), n, h, wd,
                                              cp, o, r, s, stride, pad, wsl.data_ptr(), wsl.numel() * 4, _stream()),
                        "conv2d_wgrad_acc",
                    )
                else:
                    dwp = torch.empty((op, r, s, cp), dtype=torch.float32, device=dev)
                    check(
                        L.so_conv2d_wgrad(dy.data_ptr(), _ld(dy), xr.data_ptr(), _ld(xr), dwp.data_ptr(), n, h, wd, cp, op, r, s,
                                          stride, pad, wsl.data_ptr(), wsl.numel() * 4, _stream()),
                        "conv2d_wgrad",
                    )
                    keep.append(dwp)
                    # the first o*r*s rows are the real output channels; drop the padded input channels while copying
                    if w_direct is not None:
                        check(L.so_copy2d(dwp.data_ptr(), cp, i, w_direct.grad.data_ptr(), i, i, o * r * s, 1, _stream()), "copy2d")
                    elif plain:
                        dw_ = dwp.permute(0, 3, 1, 2)
                    else:
                        dwd = torch.empty((o, r, s, i), dtype=torch.float32, device=dev)
                        check(L.so_copy2d(dwp.data_ptr(), cp, i, dwd.data_ptr(), i, i, o * r * s, 0, _stream()), "copy2d")
                        dw_ = dwd.permute(0, 3, 1, 2)
            if need_b:
                if hint is not None and op == wshape[0] and act == ACT_NONE and hint.numel() == wshape[0]:
                    if b_direct is not None:
                        check(L.so_axpby(hint.data_ptr(), 1.0, b_direct.grad.data_ptr(), 1.0, wshape[0], _stream()), "axpby")
                        keep.append(hint)
                    else:
                        db_ = hint
                elif zero_bias_grad:
                    # a bias in front of Instance/BatchNorm has an analytically zero gradient (the norm removes any
                    # per-channel constant); the reference computes round-off noise there.  Write the exact zero.
                    if b_direct is None:
                        db_ = fill_(torch.empty(o, dtype=torch.float32, device=dev), 0.0)
                elif b_direct is not None and op == o:
                    wsb = workspace(dev, L.so_colsum_ws_floats(rows_out, o) * 4, lane=lane + 2)
                    check(L.so_colsum(dy.data_ptr(), _ld(dy), rows_out, o, b_direct.grad.data_ptr(), 1, wsb.data_ptr(), _stream()), "colsum")
                else:
                    dbp = _colsum(dy.data_ptr(), _ld(dy), rows_out, op, dev)
                    if b_direct is not None:
                        check(L.so_axpby(dbp.data_ptr(), 1.0, b_direct.grad.data_ptr(), 1.0, o, _stream()), "axpby")
                        keep.append(dbp)
                    else:
                        db_ = dbp[:o]
            return dw_, db_

        if ctx.needs_input_grad[0]:
            ws = workspace(dev)
            dxp = nhwc_empty(n, h, wd, cp, dev)
            wino = _wino_mode(op, cp, n, h, wd) if (r == 3 and s == 3 and stride == 1 and pad == 1 and
                                                      n * h * wd >= WINOGRAD_TRAINABLE_MIN_PIXELS) else "direct"
            if wino != "direct":
                # the input gradient of a 3x3 / s1 / p1 convolution = the same convolution with flipped taps, C <-> Ko
                u = _wino_weights(w, None, True, fused=wino == "fused", ko_pad=op, f44=wino == "nonfused4")
                wino_conv3x3(dy.data_ptr(), _ld(dy), u, None, None, dxp.data_ptr(), cp, n, h, wd, op, cp, ACT_NONE, dev,
                             fused=wino == "fused", f44=wino == "nonfused4")
            else:
                # trainable weights change every step: read them in place (OHWI rows are contiguous along the GEMM
                # column c) instead of writing a transposed copy per step
                check(
                    L.so_conv2d_dgrad(dy.data_ptr(), _ld(dy), _pad_rows(w, op).data_ptr(), dxp.data_ptr(), cp, n, h, wd, cp, op,
                                      r, s, stride, pad, ws.data_ptr(), ws.numel() * 4, _stream()),
                    "conv2d_dgrad",
                )
            dx = dxp if cp == i else dxp[:, :i]
        if need_w or need_b:
            dw, db = weight_grads(lane=0)
        grad_ready(w_direct, b_direct)   # both the input- and the weight-gradient kernels of this layer are in the stream
        return dx, dw, db, None, None, None, None, None, None, None, None, None


def conv2d(x, weight, bias=None, stride=1, padding=1, act=ACT_NONE, zero_bias_grad=False, act_grad_external=False,
           act_param=0.0, bias_grad_hint=False, bias_grad_external=False):
    """nn.Conv2d forward (+ optional fused ReLU epilogue) on fp32 MFMA.  zero_bias_grad: the caller guarantees the
    output feeds an Instance/BatchNorm directly, so d loss / d bias is exactly zero and is not computed.
    act_grad_external: the only consumer of the output multiplies the gradient by the activation's mask itself
    (batch_norm_train(relu_gate_input=True)), so the backward pass skips its own mask kernel.
    bias_grad_external: that consumer also accumulates THIS convolution's bias gradient into the optimizer's slab
    (batch_norm_train(conv_bias=bias)); the backward pass here then computes none."""
    if bias_grad_external and not (bias is not None and _direct_grad_ok(bias, False)):
        raise RuntimeError("conv2d(bias_grad_external=True) needs a bias whose gradient lives in the optimizer's slab")
    return _Conv2dFn.apply(x, weight, bias, stride, padding, act, zero_bias_grad, act_grad_external, act_param, bias_grad_hint,
                           _forward_only(x, weight, bias), bias_grad_external)


def _forward_only(*ts):
    """True when no gradient can be asked of this call (grad mode off, or nothing requires grad).  Inside an
    autograd.Function's forward grad mode is always off and needs_input_grad follows requires_grad, so the wrappers decide."""
    return not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts))


# ------------------------------------------------------------------------------------------------
# pointwise
# ------------------------------------------------------------------------------------------------
class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, param):
        x = to_rows(x)
        n, c, h, w = x.shape
        y = nhwc_empty(n, h, w, c, x.device)
        check(lib().so_act_fwd(x.data_ptr(), _ld(x), y.data_ptr(), c, n * h * w, c, act, param, _stream()), "act_fwd")
        ctx.save_for_backward(x)
        ctx.cfg = (act, param)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        act, param = ctx.cfg
        dy = to_rows(dy)
        n, c, h, w = x.shape
        dx = nhwc_empty(n, h, w, c, x.device)
        check(lib().so_act_bwd(x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), dx.data_ptr(), c, n * h * w, c, act, param, _stream()), "act_bwd")
        return dx, None, None


def activation(x, kind, param=0.0):
    return _ActFn.apply(x, ACT_CODES[kind] if not isinstance(kind, int) else kind, float(param))


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        L = lib()
        a, b = to_rows(a), to_rows(b)
        n, c, h, w = a.shape
        out = nhwc_empty(n, h, w, c, a.device)
        check(L.so_add(a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), out.data_ptr(), c, n * h * w, c, _stream()), "add")
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    """a + b of two same-shape activations (residual connections, sams/spade.py:171)."""
    if a.shape != b.shape:
        raise ValueError(f"add: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
    return _AddFn.apply(a, b)


class _CatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *xs):
        L = lib()
        n, _, h, w = xs[0].shape
        cs = [x.shape[1] for x in xs]
        ct = sum(cs)
        # the slab's pitch is rounded up to a multiple of 4 and the pad columns are ZERO (written by the last source's kernel):
        # a convolution that needs its input channels padded (ops.conv2d: 16-byte loaders) reads the slab in place instead of
        # copying it into a padded buffer first (cat_channels marks the result `_so_zero_padded`)
        ld = (ct + 3) // 4 * 4
        out = nhwc_empty(n, h, w, ct, xs[0].device, ld=ld)
        off = 0
        for k, (x, c) in enumerate(zip(xs, cs)):
            _require_cuda(x)
            if x.dtype != torch.float32:
                raise TypeError("fp32 only")
            cd = c + (ld - ct if k == len(xs) - 1 else 0)   # the last source also writes the pad columns (zeros)
            if _is_rows(x):
                check(L.so_copy2d(x.data_ptr(), _ld(x), c, out.data_ptr() + 4 * off, ld, cd, n * h * w, 0, _stream()), "copy2d")
            else:   # planar (batch tensors): transposed straight into its channel range, no private NHWC copy first
                src = x if x.is_contiguous() else x.contiguous()
                check(L.so_nchw_to_nhwc(src.data_ptr(), out.data_ptr() + 4 * off, ld, n, c, cd, h * w, _stream()), "nchw_to_nhwc")
            off += c
        ctx.cs = cs
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = to_rows(dy)
        outs, off = [], 0
        for c in ctx.cs:
            outs.append(dy[:, off:off + c])
            off += c
        return tuple(outs)


def cat_channels(xs):
    """torch.cat(xs, dim=1) for NHWC-pitch tensors (gradient = channel slices, no copy)."""
    y = _CatFn.apply(*xs)
    if _is_rows(y) and _ld(y) != y.shape[1]:
        y._so_zero_padded = _ld(y)   # pitch = channels rounded up to 4, pad columns zero (see _CatFn.forward)
    return y


class _CatBatchFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *xs):
        L = lib()
        xs = [to_rows(x) for x in xs]
        _, c, h, w = xs[0].shape
        ns = [x.shape[0] for x in xs]
        out = nhwc_empty(sum(ns), h, w, c, xs[0].device)
        row = 0
        for x, n in zip(xs, ns):
            check(L.so_copy2d(x.data_ptr(), _ld(x), c, out.data_ptr() + 4 * row * c, c, c, n * h * w, 0, _stream()), "copy2d")
            row += n * h * w
        ctx.ns = ns
        return out

    @staticmethod
    def backward(ctx, dy):
        outs, k = [], 0
        for n in ctx.ns:
            outs.append(dy[k:k + n])
            k += n
        return tuple(outs)


def cat_batch(xs):
    """torch.cat(xs, dim=0) of NHWC-pitch activations (sams_model.py:397: fake and real halves of a discriminator batch)."""
    return _CatBatchFn.apply(*xs)


class _Upsample2xFn(torch.autograd.Function):
    """y = upsample2x(act(x)); act = ACT_NONE gives the plain nn.Upsample."""

    @staticmethod
    def forward(ctx, x, act, param):
        x = to_rows(x)
        n, c, h, w = x.shape
        y = nhwc_empty(n, 2 * h, 2 * w, c, x.device)
        check(lib().so_upsample2x_act_fwd(x.data_ptr(), _ld(x), y.data_ptr(), c, n, h, w, c, act, param, _stream()), "upsample2x_fwd")
        ctx.shape = (n, c, h, w)
        ctx.cfg = (act, param)
        if act != ACT_NONE:
            ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        act, param = ctx.cfg
        dy = to_rows(dy)
        dx = nhwc_empty(n, h, w, c, dy.device)
        x = ctx.saved_tensors[0] if act != ACT_NONE else None
        check(lib().so_upsample2x_act_bwd(x.data_ptr() if x is not None else None, _ld(x) if x is not None else 0,
                                          dy.data_ptr(), _ld(dy), dx.data_ptr(), c, n, h, w, c, act, param, _stream()), "upsample2x_bwd")
        return dx, None, None


def upsample2x_bilinear(x, act=ACT_NONE, param=0.0):
    """nn.Upsample(scale_factor=2, mode="bilinear"); with `act` the activation in front of it is applied in the same
    pass (the U-Net up path's `act -> Upsample`, unet.py:137-138)."""
    return _Upsample2xFn.apply(x, ACT_CODES[act] if not isinstance(act, int) else act, float(param))


class _Upsample2xCatFn(torch.autograd.Function):
    """y = upsample2x(act(cat([a, b], 1))) with the concatenation never materialised (U-Net skip connection feeding the
    enclosing block's `act -> Upsample`); the backward pass writes the two input gradients separately."""

    @staticmethod
    def forward(ctx, a, b, act, param):
        a, b = to_rows(a), to_rows(b)
        n, c1, h, w = a.shape
        c2 = b.shape[1]
        y = nhwc_empty(n, 2 * h, 2 * w, c1 + c2, a.device)
        check(lib().so_upsample2x_cat_fwd(a.data_ptr(), _ld(a), c1, b.data_ptr(), _ld(b), c2, y.data_ptr(), c1 + c2, n, h, w,
                                          act, param, _stream()), "upsample2x_cat_fwd")
        ctx.shape = (n, c1, c2, h, w)
        ctx.cfg = (act, param)
        if act != ACT_NONE:
            ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c1, c2, h, w = ctx.shape
        act, param = ctx.cfg
        dy = to_rows(dy)
        da, db = nhwc_empty(n, h, w, c1, dy.device), nhwc_empty(n, h, w, c2, dy.device)
        a, b = ctx.saved_tensors if act != ACT_NONE else (None, None)
        check(lib().so_upsample2x_cat_bwd(a.data_ptr() if a is not None else None, _ld(a) if a is not None else 0, c1,
                                          b.data_ptr() if b is not None else None, _ld(b) if b is not None else 0, c2,
                                          dy.data_ptr(), _ld(dy), da.data_ptr(), c1, db.data_ptr(), c2, n, h, w, act, param,
                                          _stream()), "upsample2x_cat_bwd")
        return da, db, None, None


def upsample2x_bilinear_cat(a, b, act=ACT_NONE, param=0.0):
    return _Upsample2xCatFn.apply(a, b, ACT_CODES[act] if not isinstance(act, int) else act, float(param))


class _MaxPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = to_rows(x)
        n, c, h, w = x.shape
        y = nhwc_empty(n, h // 2, w // 2, c, x.device)
        check(lib().so_maxpool2_fwd(x.data_ptr(), _ld(x), y.data_ptr(), c, n, h, w, c, _stream()), "maxpool2_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = to_rows(dy)
        dx = nhwc_empty(n, h, w, c, x.device)
        check(lib().so_maxpool2_bwd(x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), dx.data_ptr(), c, n, h, w, c, 0, _stream()), "maxpool2_bwd")
        return dx


def maxpool2x2(x):
    return _MaxPool2Fn.apply(x)


# ------------------------------------------------------------------------------------------------
# normalisation
# ------------------------------------------------------------------------------------------------
class _NormFn(torch.autograd.Function):
    """InstanceNorm2d (instance=True) or BatchNorm2d training (instance=False)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, instance, momentum, eps, relu_gate_input=False, act2=None,
                act2_param=0.0, conv_bias=None):
        """act2 (an activation code): also return act2(y), written by the same launch - the consumer's first operation; that
        second output carries no gradient here (ops.instance_norm_act hangs the activation's own backward node on it)."""
        L = lib()
        ctx.set_materialize_grads(False)   # no zero-filled gradient tensor (an ATen fill launch) for the second output
        x = to_rows(x)
        n, c, h, w = x.shape
        G, R = (n, h * w) if instance else (1, n * h * w)
        y = nhwc_empty(n, h, w, c, x.device)
        a = nhwc_empty(n, h, w, c, x.device) if act2 is not None else None
        mean = torch.empty((G, c), dtype=torch.float32, device=x.device)
        rstd = torch.empty((G, c), dtype=torch.float32, device=x.device)
        ws = workspace(x.device, L.so_norm_ws_floats(G, R, c) * 4)
        tail = (eps, gamma.data_ptr() if gamma is not None else None, beta.data_ptr() if beta is not None else None,
                mean.data_ptr(), rstd.data_ptr(),
                running_mean.data_ptr() if running_mean is not None else None,
                running_var.data_ptr() if running_var is not None else None,
                momentum, ws.data_ptr(), _stream())
        if a is None:
            check(L.so_norm_fwd(x.data_ptr(), _ld(x), y.data_ptr(), c, G, R, c, *tail), "norm_fwd")
        else:
            check(L.so_norm_act_fwd(x.data_ptr(), _ld(x), y.data_ptr(), c, a.data_ptr(), c, int(act2), float(act2_param), G, R, c,
                                    *tail), "norm_act_fwd")
        ctx.save_for_backward(x, mean, rstd, gamma)
        ctx.cfg = (G, R, bool(relu_gate_input))
        # Conv -> ReLU -> BatchNorm: the bias gradient of that convolution = the column sums of the gated gradient this node
        # returns; with the bias living in the optimizer's slab the backward pass accumulates it there (so_norm_bwd_bias)
        if conv_bias is not None and not (relu_gate_input and not instance and _direct_grad_ok(conv_bias, False)
                                          and conv_bias.numel() == c):
            raise RuntimeError("batch_norm_train(conv_bias=...) needs relu_gate_input, batch statistics and a bias whose "
                               "gradient lives in the optimizer's slab (the producing convolution skips its own bias gradient)")
        ctx.conv_bias = conv_bias
        # affine parameters living in the optimizer's flat slab get their gradients accumulated in place
        ctx.direct = (gamma, beta) if gamma is not None and _direct_grad_ok(gamma, False) and _direct_grad_ok(beta, False) else None
        if a is None:
            return y
        ctx.mark_non_differentiable(a)
        return y, a

    @staticmethod
    def backward(ctx, dy, _da=None):
        L = lib()
        if dy is None:
            return (None,) * 12
        x, mean, rstd, gamma = ctx.saved_tensors
        G, R, relu_gate = ctx.cfg
        n, c, h, w = x.shape
        dy = to_rows(dy)
        dx = nhwc_empty(n, h, w, c, x.device)
        dgamma = dbeta = None
        gptr = bptr = None
        if ctx.direct is not None:
            gptr, bptr = ctx.direct[0].grad.data_ptr(), ctx.direct[1].grad.data_ptr()
        elif gamma is not None:
            dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
            dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
            gptr, bptr = dgamma.data_ptr(), dbeta.data_ptr()
        ws = workspace(x.device, L.so_norm_ws_floats(G, R, c) * 4)
        head = (x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), dx.data_ptr(), c, G, R, c, mean.data_ptr(), rstd.data_ptr(),
                gamma.data_ptr() if gamma is not None else None, gptr, bptr, int(ctx.direct is not None), int(relu_gate))
        if ctx.conv_bias is not None:
            check(L.so_norm_bwd_bias(*head, ctx.conv_bias.grad.data_ptr(), 1, ws.data_ptr(), _stream()), "norm_bwd_bias")
        else:
            check(L.so_norm_bwd(*head, ws.data_ptr(), _stream()), "norm_bwd")
        if ctx.direct is not None:
            grad_ready(*ctx.direct)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None


def instance_norm(x, eps=1e-5):
    return _NormFn.apply(x, None, None, None, None, True, 0.0, eps)


class _ForkActFn(torch.autograd.Function):
    """(x, act(x)) for a tensor that feeds BOTH a skip connection and an activation - every U-Net block input
    (models/networks/cpvton/unet.py:187-198: `torch.cat([x, self.model(x)], 1)` with `self.model` starting with the block's
    down activation).  Left to autograd that is act_bwd, then an accumulation add of the two gradients (an ATen launch);
    here the backward pass is ONE kernel, dx = d_skip + d_act * act'(x), with the same roundings in the same order
    (so_act_bwd_add), i.e. the same bits.  `a` given: the activation's VALUE was already written by the kernel that
    produced x (so_norm_act_fwd) and the forward launches nothing."""

    @staticmethod
    def forward(ctx, x, a, act, param):
        ctx.set_materialize_grads(False)
        x = to_rows(x)
        n, c, h, w = x.shape
        if a is None:
            a = nhwc_empty(n, h, w, c, x.device)
            check(lib().so_act_fwd(x.data_ptr(), _ld(x), a.data_ptr(), c, n * h * w, c, act, param, _stream()), "act_fwd")
        else:
            a = a.detach()
        ctx.save_for_backward(x)
        ctx.cfg = (act, param)
        return x.view_as(x), a

    @staticmethod
    def backward(ctx, d_skip, d_act):
        (x,) = ctx.saved_tensors
        act, param = ctx.cfg
        if d_act is None:
            return d_skip, None, None, None
        n, c, h, w = x.shape
        d_act = to_rows(d_act)
        dx = nhwc_empty(n, h, w, c, x.device)
        if d_skip is None:
            check(lib().so_act_bwd(x.data_ptr(), _ld(x), d_act.data_ptr(), _ld(d_act), dx.data_ptr(), c, n * h * w, c, act, param,
                                   _stream()), "act_bwd")
        else:
            d_skip = to_rows(d_skip)
            check(lib().so_act_bwd_add(x.data_ptr(), _ld(x), d_act.data_ptr(), _ld(d_act), d_skip.data_ptr(), _ld(d_skip),
                                       dx.data_ptr(), c, n * h * w, c, act, param, _stream()), "act_bwd_add")
        return dx, None, None, None


def fork_act(x, kind, param=0.0, precomputed=None):
    """(x_skip, act(x)): x_skip is x itself, routed through the node whose backward merges the skip gradient with the
    activation's (see _ForkActFn)."""
    code = ACT_CODES[kind] if not isinstance(kind, int) else kind
    return _ForkActFn.apply(x, precomputed, code, float(param))


def instance_norm_act(x, eps, kind, param=0.0):
    """(y, act(y)) with y = instance_norm(x): one launch for both - the U-Net hands y to the skip connection and act(y) to
    the next block's convolution (models/networks/cpvton/unet.py:132-147, 187); one backward kernel for the pair."""
    code = ACT_CODES[kind] if not isinstance(kind, int) else kind
    y, a = _NormFn.apply(x, None, None, None, None, True, 0.0, eps, False, code, float(param))
    return _ForkActFn.apply(y, a, code, float(param))


def batch_norm_train(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, relu_gate_input=False, conv_bias=None):
    """relu_gate_input: x is the output of a ReLU whose producer skips its own backward mask (conv2d(...,
    act_grad_external=True)); the returned input gradient is then the one in front of that ReLU.  conv_bias: the bias
    parameter of that producing convolution - its gradient is accumulated by this node's backward pass (so_norm_bwd_bias)
    when it lives in the optimizer's slab, and the convolution's own backward then skips its column sums."""
    return _NormFn.apply(x, gamma, beta, running_mean, running_var, False, momentum, eps, relu_gate_input, None, 0.0, conv_bias)


def batch_norm_eval(x, gamma, beta, running_mean, running_var, eps=1e-5):
    x = to_rows(x)
    n, c, h, w = x.shape
    y = nhwc_empty(n, h, w, c, x.device)
    check(
        lib().so_norm_apply(x.data_ptr(), _ld(x), y.data_ptr(), c, 1, n * h * w, c, running_mean.data_ptr(),
                            running_var.data_ptr(), 1, eps, gamma.data_ptr() if gamma is not None else None,
                            beta.data_ptr() if beta is not None else None, _stream()),
        "norm_apply",
    )
    return y


# ------------------------------------------------------------------------------------------------
# SAGAN self-attention (models/networks/attention/sagan.py:29-54)
# ------------------------------------------------------------------------------------------------
def _gemm(ta, tb, M, N, K, A, lda, sa, B, ldb, sb, C, ldc, sc, batch, alpha=None, bias=None, res=None, ldres=0, sres=0,
          act=ACT_NONE, device=None):
    ws = workspace(device)
    check(
        lib().so_gemm_batched(
            ta, tb, M, N, K, A, lda, sa, B, ldb, sb, C, ldc, sc, batch,
            alpha, bias, res, ldres, sres, act, 0.0, ws.data_ptr(), ws.numel() * 4, _stream(),
        ),
        "gemm_batched",
    )


def _adjacent(ts):
    """True if the dense tensors `ts` follow each other in memory without gaps (views of one flat slab)."""
    return all(t.is_contiguous() or t.permute(0, 2, 3, 1).is_contiguous() if t.dim() == 4 else t.is_contiguous() for t in ts) and \
        all(a.data_ptr() + a.numel() * 4 == b.data_ptr() for a, b in zip(ts[:-1], ts[1:]))


class _SelfAttentionQkvFn(torch.autograd.Function):
    """Self-attention when optim.HipAdam has laid the q/k/v weights, biases and their gradients out adjacently: the three
    1x1 projections are ONE GEMM into a [rows][2d+C] buffer (q, k, v = column slices), and in the backward pass the input
    gradient, the weight gradient and the bias gradient of all three are one GEMM / one column sum each, accumulated
    straight into the gradient slab.  15 launches per module instead of 24 (each costs >= 5 us inside the graph)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, gamma):
        L = lib()
        x = _dense_rows(x)
        b, c, h, w = x.shape
        n, d = h * w, wq.shape[0]
        E = 2 * d + c
        dev = x.device
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        ldx, xp = _ld(x), x.data_ptr()
        qkv = f(b * n, E)
        _gemm(0, 1, b * n, E, c, xp, ldx, 0, wq.data_ptr(), c, 0, qkv.data_ptr(), E, 0, 1, bias=bq.data_ptr(), device=dev)
        qp, kp, vp = qkv.data_ptr(), qkv.data_ptr() + d * 4, qkv.data_ptr() + 2 * d * 4
        a, o = f(b * n, n), f(b * n, c)
        out = nhwc_empty(b, h, w, c, dev)
        # energy -> softmax -> attention x V -> gamma * o + x as batched GEMMs + row softmax (an LDS-resident single-kernel core
        # was built and measured slower at B = 4 - 16-96 blocks of dependent MFMA chains against launches already at the
        # ~10 us floor - and removed in round 3)
        e = f(b * n, n)
        _gemm(0, 1, n, n, d, qp, E, n * E, kp, E, n * E, e.data_ptr(), n, n * n, b, device=dev)
        check(L.so_softmax_rows_fwd(e.data_ptr(), n, a.data_ptr(), n, b * n, n, _stream()), "softmax_fwd")
        _gemm(0, 0, n, c, n, a.data_ptr(), n, n * n, vp, E, n * E, o.data_ptr(), c, n * c, b, device=dev)
        check(L.so_scale_add(o.data_ptr(), c, gamma.data_ptr(), xp, ldx, out.data_ptr(), c, b * n, c, _stream()), "scale_add")
        ctx.save_for_backward(x, qkv, a, o, gamma)
        ctx.params = (wq, bq, gamma)  # first tensors of the adjacent weight / bias runs, and gamma
        ctx.ready = (wq, bq, wk, bk, wv, bv, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = lib()
        x, qkv, a, o, gamma = ctx.saved_tensors
        wq, bq, gpar = ctx.params
        b, c, h, w = x.shape
        n = h * w
        E = qkv.shape[1]
        d = (E - c) // 2
        dev = x.device
        dout = _dense_rows(dout)
        ldg, gp = _ld(dout), dout.data_ptr()
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        ws = workspace(dev)
        check(L.so_dot(gp, ldg, o.data_ptr(), c, b * n, c, 1.0, gpar.grad.data_ptr(), 1, ws.data_ptr(), _stream()), "dot")
        gm = gamma.data_ptr()
        qp, kp, vp = qkv.data_ptr(), qkv.data_ptr() + d * 4, qkv.data_ptr() + 2 * d * 4
        dqkv = f(b * n, E)
        de = f(b * n, n)
        dqp, dkp, dvp = dqkv.data_ptr(), dqkv.data_ptr() + d * 4, dqkv.data_ptr() + 2 * d * 4
        _gemm(1, 0, n, c, n, a.data_ptr(), n, n * n, gp, ldg, n * ldg, dvp, E, n * E, b, alpha=gm, device=dev)
        da = f(b * n, n)
        _gemm(0, 1, n, n, c, gp, ldg, n * ldg, vp, E, n * E, da.data_ptr(), n, n * n, b, alpha=gm, device=dev)
        check(L.so_softmax_rows_bwd(a.data_ptr(), n, da.data_ptr(), n, de.data_ptr(), n, b * n, n, _stream()), "softmax_bwd")
        _gemm(0, 0, n, d, n, de.data_ptr(), n, n * n, kp, E, n * E, dqp, E, n * E, b, device=dev)
        _gemm(1, 0, n, d, n, de.data_ptr(), n, n * n, qp, E, n * E, dkp, E, n * E, b, device=dev)
        # dx = dout + [dq | dk | dv] [Wq; Wk; Wv]
        dx = nhwc_empty(b, h, w, c, dev)
        _gemm(0, 0, b * n, c, E, dqkv.data_ptr(), E, 0, wq.data_ptr(), c, 0, dx.data_ptr(), c, 0, 1, res=gp, ldres=ldg, device=dev)
        # [dWq; dWk; dWv] += [dq | dk | dv]^T x   and   [dbq; dbk; dbv] += column sums, in place in the gradient slab
        wg = wq.grad.data_ptr()
        _gemm(1, 0, E, c, b * n, dqkv.data_ptr(), E, 0, x.data_ptr(), _ld(x), 0, wg, c, 0, 1, res=wg, ldres=c, device=dev)
        wsb = workspace(dev, L.so_colsum_ws_floats(b * n, E) * 4, lane=2)
        check(L.so_colsum(dqkv.data_ptr(), E, b * n, E, bq.grad.data_ptr(), 1, wsb.data_ptr(), _stream()), "colsum")
        grad_ready(*ctx.ready)
        return dx, None, None, None, None, None, None, None


class _SelfAttentionCoreFn(torch.autograd.Function):
    """_SelfAttentionQkvFn's layout (one projection GEMM into [rows][2d + C]; input, weight and bias gradients of the three
    projections as one GEMM / one column sum each, accumulated straight into the gradient slab) with everything between the
    projections in csrc/attn.hip: energy -> softmax -> attention x V -> gamma * o + x is ONE launch, its backward pass three
    (so_attn_bwd's two + so_attn_tail, which also finishes d gamma) - 8 launches per module forward + backward instead of the
    composed form's 15 + split-K reduces, each of which costs >= 5 us as a graph node."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, gamma):
        L = lib()
        x = _dense_rows(x)
        b, c, h, w = x.shape
        n, d = h * w, wq.shape[0]
        E = 2 * d + c
        dev = x.device
        ldx, xp = _ld(x), x.data_ptr()
        qkv = torch.empty((b * n, E), dtype=torch.float32, device=dev)
        _gemm(0, 1, b * n, E, c, xp, ldx, 0, wq.data_ptr(), c, 0, qkv.data_ptr(), E, 0, 1, bias=bq.data_ptr(), device=dev)
        a = torch.empty((b, n, n), dtype=torch.float32, device=dev)
        o = torch.empty((b * n, c), dtype=torch.float32, device=dev)
        out = nhwc_empty(b, h, w, c, dev)
        check(L.so_attn_fwd(qkv.data_ptr(), E, d, xp, ldx, gamma.data_ptr(), a.data_ptr(), o.data_ptr(), out.data_ptr(), c,
                            b, n, c, _stream()), "attn_fwd")
        ctx.save_for_backward(x, qkv, a, o, gamma)
        ctx.params = (wq, bq, gamma)  # first tensors of the adjacent weight / bias runs, and gamma
        ctx.ready = (wq, bq, wk, bk, wv, bv, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = lib()
        x, qkv, a, o, gamma = ctx.saved_tensors
        wq, bq, gpar = ctx.params
        b, c, h, w = x.shape
        n = h * w
        E = qkv.shape[1]
        d = (E - c) // 2
        dev = x.device
        dout = _dense_rows(dout)
        ldg, gp = _ld(dout), dout.data_ptr()
        dqkv = torch.empty((b * n, E), dtype=torch.float32, device=dev)
        scratch = torch.empty(L.so_attn_ws_floats(b, n), dtype=torch.float32, device=dev)   # dE + the fp64 d gamma partials
        check(L.so_attn_bwd(qkv.data_ptr(), E, d, a.data_ptr(), o.data_ptr(), gp, ldg, gamma.data_ptr(), dqkv.data_ptr(),
                            scratch.data_ptr(), b, n, c, _stream()), "attn_bwd")
        # dx = dout + [dq | dk | dv] [Wq; Wk; Wv]
        dx = nhwc_empty(b, h, w, c, dev)
        _gemm(0, 0, b * n, c, E, dqkv.data_ptr(), E, 0, wq.data_ptr(), c, 0, dx.data_ptr(), c, 0, 1, res=gp, ldres=ldg, device=dev)
        # [dWq; dWk; dWv] += [dq | dk | dv]^T x, in place in the gradient slab
        wg = wq.grad.data_ptr()
        _gemm(1, 0, E, c, b * n, dqkv.data_ptr(), E, 0, x.data_ptr(), _ld(x), 0, wg, c, 0, 1, res=wg, ldres=c, device=dev)
        # [dbq; dbk; dbv] += column sums of dqkv;  d gamma += <dout, o>
        check(L.so_attn_tail(dqkv.data_ptr(), E, scratch.data_ptr(), b, n, bq.grad.data_ptr(), 1, gpar.grad.data_ptr(), 1,
                             _stream()), "attn_tail")
        grad_ready(*ctx.ready)
        return dx, None, None, None, None, None, None, None


# tools / tests: False sends every shape through the composed engine launches (A/B of csrc/attn.hip against them)
ATTN_CORE = True

_QKV_CACHE = {}


def _stacked_qkv(wq, bq, wk, bk, wv, bv):
    """([Wq; Wk; Wv] as one [2d + C][C] matrix, [bq; bk; bv]) for a forward pass that needs no gradients, or None.  The three
    1x1 projections then run as ONE GEMM (4 launches fewer per module; each costs >= 4 us inside a replayed graph).  Copies are
    cached per parameter objects and versions.  Parameters that are views of a larger storage (planted in an optimizer's flat
    slab, which is updated through raw pointers without a version bump) are never cached: adjacent ones are used in place,
    others take the three-GEMM path."""
    ws_, bs_ = (wq, wk, wv), (bq, bk, bv)
    if any(t.shape[0] % 4 for t in ws_):
        return None
    if _adjacent(ws_) and _adjacent(bs_) and all(t.dim() != 4 or t.permute(0, 2, 3, 1).is_contiguous() for t in ws_):
        return wq, bq
    if any(t.untyped_storage().nbytes() != t.numel() * 4 for t in ws_ + bs_):
        return None
    key = tuple(id(t) for t in ws_ + bs_)
    ver = tuple((t._version, t.data_ptr()) for t in ws_ + bs_)
    hit = _QKV_CACHE.get(key)
    if hit is not None and all(r() is t for r, t in zip(hit[0], ws_ + bs_)) and hit[1] == ver:
        return _serve_cached(hit[2], hit[3])
    if torch.cuda.is_current_stream_capturing():
        return None    # no entry is created under capture (the stacked copy would be recorded, not run): the three-GEMM path
    c = wq.shape[1]
    with torch.no_grad():
        wcat = torch.cat([_ohwi(t).reshape(t.shape[0], c) for t in ws_], 0).contiguous()
        bcat = torch.cat([t.detach() for t in bs_], 0).contiguous()
    for k in [k for k, v in _QKV_CACHE.items() if any(r() is None for r in v[0])]:
        del _QKV_CACHE[k]
    _QKV_CACHE[key] = (tuple(weakref.ref(t) for t in ws_ + bs_), ver, wcat, bcat)
    return wcat, bcat


class _SelfAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, gamma):
        L = lib()
        x = _dense_rows(x)
        b, c, h, w = x.shape
        n = h * w
        d = wq.shape[0]
        dev = x.device
        if c % 4 or n % 4:
            raise RuntimeError("self-attention kernels need C and H*W to be multiples of 4")
        wq2, wk2, wv2 = (_ohwi(t).reshape(t.shape[0], c) for t in (wq, wk, wv))
        d_true = d
        if d % 4:  # e.g. C = 1336 -> d = 167: zero-pad the query/key projections to a multiple of 4 (adds 0 to Q.K)
            d = (d + 3) // 4 * 4
            wq2, wk2 = (_pad_rows(t.reshape(d_true, 1, 1, c), d).reshape(d, c) for t in (wq2, wk2))
            pad_vec = lambda t: torch.cat([t.detach(), torch.zeros(d - d_true, dtype=torch.float32, device=dev)])
            bq, bk = pad_vec(bq), pad_vec(bk)
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        q, k, v = f(b * n, d), f(b * n, d), f(b * n, c)
        ldx = _ld(x)
        xp = x.data_ptr()
        # 1x1 convolutions: rows x C  @  W^T  (+ bias)
        _gemm(0, 1, b * n, d, c, xp, ldx, 0, wq2.data_ptr(), c, 0, q.data_ptr(), d, 0, 1, bias=bq.data_ptr(), device=dev)
        _gemm(0, 1, b * n, d, c, xp, ldx, 0, wk2.data_ptr(), c, 0, k.data_ptr(), d, 0, 1, bias=bk.data_ptr(), device=dev)
        _gemm(0, 1, b * n, c, c, xp, ldx, 0, wv2.data_ptr(), c, 0, v.data_ptr(), c, 0, 1, bias=bv.data_ptr(), device=dev)
        # energy[b] = q[b] k[b]^T ; attention = softmax(rows)
        e = f(b * n, n)
        _gemm(0, 1, n, n, d, q.data_ptr(), d, n * d, k.data_ptr(), d, n * d, e.data_ptr(), n, n * n, b, device=dev)
        a = f(b * n, n)
        check(L.so_softmax_rows_fwd(e.data_ptr(), n, a.data_ptr(), n, b * n, n, _stream()), "softmax_fwd")
        # o[b] = attention[b] v[b] ; out = gamma * o + x
        o = f(b * n, c)
        _gemm(0, 0, n, c, n, a.data_ptr(), n, n * n, v.data_ptr(), c, n * c, o.data_ptr(), c, n * c, b, device=dev)
        out = nhwc_empty(b, h, w, c, dev)
        check(L.so_scale_add(o.data_ptr(), c, gamma.data_ptr(), xp, ldx, out.data_ptr(), c, b * n, c, _stream()), "scale_add")
        ctx.save_for_backward(x, wq2, wk2, wv2, q, k, v, a, o, gamma)
        ctx.shapes = (tuple(wq.shape), tuple(wk.shape), tuple(wv.shape), d_true)
        # parameters planted in the optimizer's flat slab: their gradients are accumulated in place by the kernels
        params = (wq, bq, wk, bk, wv, bv, gamma)
        ok = d == d_true and all(_direct_grad_ok(t, ohwi=t.dim() == 4) for t in params)
        ctx.direct = params if ok else None
        return out

    @staticmethod
    def backward(ctx, dout):
        L = lib()
        x, wq2, wk2, wv2, q, k, v, a, o, gamma = ctx.saved_tensors
        b, c, h, w = x.shape
        n = h * w
        d = wq2.shape[0]
        dev = x.device
        dout = _dense_rows(dout)
        ldg = _ld(dout)
        gp = dout.data_ptr()
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        ws = workspace(dev)
        # d gamma = <dout, o>
        direct = ctx.direct
        dgamma = f(1) if direct is None else None
        check(L.so_dot(gp, ldg, o.data_ptr(), c, b * n, c, 1.0, (dgamma if direct is None else direct[6].grad).data_ptr(),
                       int(direct is not None), ws.data_ptr(), _stream()), "dot")
        gm = gamma.data_ptr()
        # dv[b] = gamma * a[b]^T dout[b] ; da[b] = gamma * dout[b] v[b]^T
        dv = f(b * n, c)
        _gemm(1, 0, n, c, n, a.data_ptr(), n, n * n, gp, ldg, n * ldg, dv.data_ptr(), c, n * c, b, alpha=gm, device=dev)
        da = f(b * n, n)
        _gemm(0, 1, n, n, c, gp, ldg, n * ldg, v.data_ptr(), c, n * c, da.data_ptr(), n, n * n, b, alpha=gm, device=dev)
        de = f(b * n, n)
        check(L.so_softmax_rows_bwd(a.data_ptr(), n, da.data_ptr(), n, de.data_ptr(), n, b * n, n, _stream()), "softmax_bwd")
        # dq[b] = de[b] k[b] ; dk[b] = de[b]^T q[b]
        dq, dk = f(b * n, d), f(b * n, d)
        _gemm(0, 0, n, d, n, de.data_ptr(), n, n * n, k.data_ptr(), d, n * d, dq.data_ptr(), d, n * d, b, device=dev)
        _gemm(1, 0, n, d, n, de.data_ptr(), n, n * n, q.data_ptr(), d, n * d, dk.data_ptr(), d, n * d, b, device=dev)
        # dx = dout + dq Wq + dk Wk + dv Wv   (1x1-conv input gradients chained through the residual epilogue)
        dx = nhwc_empty(b, h, w, c, dev)
        dxp = dx.data_ptr()
        _gemm(0, 0, b * n, c, d, dq.data_ptr(), d, 0, wq2.data_ptr(), c, 0, dxp, c, 0, 1, res=gp, ldres=ldg, device=dev)
        _gemm(0, 0, b * n, c, d, dk.data_ptr(), d, 0, wk2.data_ptr(), c, 0, dxp, c, 0, 1, res=dxp, ldres=c, device=dev)
        _gemm(0, 0, b * n, c, c, dv.data_ptr(), c, 0, wv2.data_ptr(), c, 0, dxp, c, 0, 1, res=dxp, ldres=c, device=dev)
        # weight / bias gradients: dW = dY^T X
        xp, ldx = x.data_ptr(), _ld(x)
        if direct is not None:
            for g_, ldg_, rows_, wpar, bpar in ((dq, d, d, direct[0], direct[1]), (dk, d, d, direct[2], direct[3]),
                                                (dv, c, c, direct[4], direct[5])):
                wg = wpar.grad.data_ptr()
                _gemm(1, 0, rows_, c, b * n, g_.data_ptr(), ldg_, 0, xp, ldx, 0, wg, c, 0, 1, res=wg, ldres=c, device=dev)
                wsb = workspace(dev, L.so_colsum_ws_floats(b * n, rows_) * 4, lane=2)
                check(L.so_colsum(g_.data_ptr(), ldg_, b * n, rows_, bpar.grad.data_ptr(), 1, wsb.data_ptr(), _stream()), "colsum")
            grad_ready(*direct)
            return dx, None, None, None, None, None, None, None
        dwq, dwk, dwv = f(d, c), f(d, c), f(c, c)
        _gemm(1, 0, d, c, b * n, dq.data_ptr(), d, 0, xp, ldx, 0, dwq.data_ptr(), c, 0, 1, device=dev)
        _gemm(1, 0, d, c, b * n, dk.data_ptr(), d, 0, xp, ldx, 0, dwk.data_ptr(), c, 0, 1, device=dev)
        _gemm(1, 0, c, c, b * n, dv.data_ptr(), c, 0, xp, ldx, 0, dwv.data_ptr(), c, 0, 1, device=dev)
        dbq = _colsum(dq.data_ptr(), d, b * n, d, dev)
        dbk = _colsum(dk.data_ptr(), d, b * n, d, dev)
        dbv = _colsum(dv.data_ptr(), c, b * n, c, dev)
        sq, sk, sv, d_true = ctx.shapes
        return (dx, dwq[:d_true].view(sq), dbq[:d_true], dwk[:d_true].view(sk), dbk[:d_true], dwv.view(sv), dbv, dgamma)


@torch.no_grad()
def _self_attention_forward_only(x, wcat, bcat, gamma, d):
    """q | k | v = column slices of ONE projection GEMM (as _SelfAttentionQkvFn.forward), nothing saved."""
    L = lib()
    x = _dense_rows(x)
    b, c, h, w = x.shape
    n, E, dev = h * w, 2 * d + c, x.device
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    ldx, xp = _ld(x), x.data_ptr()
    qkv = f(b * n, E)
    _gemm(0, 1, b * n, E, c, xp, ldx, 0, wcat.data_ptr(), c, 0, qkv.data_ptr(), E, 0, 1, bias=bcat.data_ptr(), device=dev)
    qp, kp, vp = qkv.data_ptr(), qkv.data_ptr() + d * 4, qkv.data_ptr() + 2 * d * 4
    out = nhwc_empty(b, h, w, c, dev)
    if ATTN_CORE and L.so_attn_supported(b, n, c, d):   # one launch for everything behind the projection (csrc/attn.hip)
        check(L.so_attn_fwd(qkv.data_ptr(), E, d, xp, ldx, gamma.data_ptr(), None, None, out.data_ptr(), c, b, n, c, _stream()),
              "attn_fwd")
        return out
    e, a = f(b * n, n), f(b * n, n)
    _gemm(0, 1, n, n, d, qp, E, n * E, kp, E, n * E, e.data_ptr(), n, n * n, b, device=dev)
    check(L.so_softmax_rows_fwd(e.data_ptr(), n, a.data_ptr(), n, b * n, n, _stream()), "softmax_fwd")
    # gamma * (attention x V) + x in the product's epilogue (alpha, residual): o itself is only needed by d gamma
    _gemm(0, 0, n, c, n, a.data_ptr(), n, n * n, vp, E, n * E, out.data_ptr(), c, n * c, b, alpha=gamma.data_ptr(), res=xp,
          ldres=ldx, sres=n * ldx, device=dev)
    return out


def self_attention(x, wq, bq, wk, bk, wv, bv, gamma):
    if x.is_cuda and x.shape[1] % 4 == 0 and (x.shape[2] * x.shape[3]) % 4 == 0 and _forward_only(x, wq, bq, wk, bk, wv, bv, gamma):
        stacked = _stacked_qkv(wq, bq, wk, bk, wv, bv)
        if stacked is not None:
            return _self_attention_forward_only(x, stacked[0], stacked[1], gamma, wq.shape[0])
    ws_, bs_ = (wq, wk, wv), (bq, bk, bv)
    # q/k/v weights, biases and their gradients adjacent in the optimizer's slab: one GEMM serves the three projections
    slab = (x.is_cuda and wq.shape[0] % 4 == 0 and x.shape[1] % 4 == 0
            and all(_direct_grad_ok(t, ohwi=True) for t in ws_) and all(_direct_grad_ok(t, ohwi=False) for t in bs_)
            and _direct_grad_ok(gamma, ohwi=False) and _adjacent(ws_) and _adjacent(bs_)
            and _adjacent([t.grad for t in ws_]) and _adjacent([t.grad for t in bs_]))
    n = x.shape[2] * x.shape[3]
    if slab and ATTN_CORE and lib().so_attn_supported(x.shape[0], n, x.shape[1], wq.shape[0]):
        fn = _SelfAttentionCoreFn      # n <= 256 positions (any n, also ragged): the fused core of csrc/attn.hip
    else:
        fn = _SelfAttentionQkvFn if slab and n % 4 == 0 else _SelfAttentionFn
    return fn.apply(x, wq, bq, wk, bk, wv, bv, gamma)


# ------------------------------------------------------------------------------------------------
# GMM pieces
# ------------------------------------------------------------------------------------------------
class _L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, transpose_hw):
        x = to_rows(x)
        n, c, h, w = x.shape
        y = nhwc_empty(n, w, h, c, x.device) if transpose_hw else nhwc_empty(n, h, w, c, x.device)
        inv = torch.empty(n * h * w, dtype=torch.float32, device=x.device)
        check(lib().so_l2norm_fwd(x.data_ptr(), _ld(x), y.data_ptr(), c, inv.data_ptr(), n, h, w, c, int(transpose_hw), _stream()), "l2norm_fwd")
        ctx.save_for_backward(y, inv)
        ctx.cfg = (n, c, h, w, transpose_hw)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        n, c, h, w, tr = ctx.cfg
        dy = to_rows(dy)
        dx = nhwc_empty(n, h, w, c, y.device)
        check(lib().so_l2norm_bwd(y.data_ptr(), c, dy.data_ptr(), _ld(dy), inv.data_ptr(), dx.data_ptr(), c, n, h, w, c, int(tr), _stream()), "l2norm_bwd")
        return dx, None


def feature_l2norm(x, transpose_hw=False):
    """FeatureL2Norm; transpose_hw=True additionally returns the (N, C, W, H) transposed map that
    FeatureCorrelation builds from feature A (warp.py:60)."""
    return _L2NormFn.apply(x, transpose_hw)


class _CorrelationFn(torch.autograd.Function):
    """corr[b, ja*h+ia, i, j] = sum_c A[b,c,ia,ja] B[b,c,i,j]; `a_t` is A already transposed to (N,C,W,H)."""

    @staticmethod
    def forward(ctx, a_t, fb):
        a_t, fb = _dense_rows(a_t), _dense_rows(fb)
        b, c, h, w = fb.shape
        n = h * w
        out = nhwc_empty(b, h, w, n, fb.device)
        _gemm(0, 1, n, n, c, fb.data_ptr(), _ld(fb), n * _ld(fb), a_t.data_ptr(), _ld(a_t), n * _ld(a_t),
              out.data_ptr(), n, n * n, b, device=fb.device)
        ctx.save_for_backward(a_t, fb)
        return out

    @staticmethod
    def backward(ctx, dy):
        a_t, fb = ctx.saved_tensors
        b, c, h, w = fb.shape
        n = h * w
        dy = _dense_rows(dy)
        ldg = _ld(dy)
        dev = fb.device
        dfb = nhwc_empty(b, h, w, c, dev)
        da = nhwc_empty(b, w, h, c, dev)
        # dB[b] = dcorr[b] A_t[b]   ;   dA_t[b] = dcorr[b]^T B[b]
        _gemm(0, 0, n, c, n, dy.data_ptr(), ldg, n * ldg, a_t.data_ptr(), _ld(a_t), n * _ld(a_t), dfb.data_ptr(), c, n * c, b, device=dev)
        _gemm(1, 0, n, c, n, dy.data_ptr(), ldg, n * ldg, fb.data_ptr(), _ld(fb), n * _ld(fb), da.data_ptr(), c, n * c, b, device=dev)
        return da, dfb


def feature_correlation(a_transposed, fb):
    return _CorrelationFn.apply(a_transposed, fb)


class _LinearTanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, apply_tanh):
        x = to_rows(x)
        n, c, h, w = x.shape
        j = weight.shape[0]
        wt = weight.detach().contiguous()
        y = torch.empty((n, j), dtype=torch.float32, device=x.device)
        check(lib().so_linear_chw_fwd(x.data_ptr(), _ld(x), wt.data_ptr(), bias.data_ptr(), y.data_ptr(), n, h * w, c, j, int(apply_tanh), _stream()), "linear_fwd")
        ctx.save_for_backward(x, wt, y)
        ctx.apply_tanh = apply_tanh
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wt, y = ctx.saved_tensors
        n, c, h, w = x.shape
        j = wt.shape[0]
        dy = dy.contiguous()
        dx = nhwc_empty(n, h, w, c, x.device) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(wt)
        db = torch.empty(j, dtype=torch.float32, device=x.device)
        check(
            lib().so_linear_chw_bwd(x.data_ptr(), _ld(x), wt.data_ptr(), y.data_ptr(), dy.data_ptr(),
                                    dx.data_ptr() if dx is not None else None, c, dw.data_ptr(), db.data_ptr(),
                                    n, h * w, c, j, int(ctx.apply_tanh), _stream()),
            "linear_bwd",
        )
        return dx, dw, db, None


def linear_chw_tanh(x, weight, bias, apply_tanh=True):
    """x.view(N, -1) in the reference's (C,H,W) order -> Linear -> tanh (warp.py:94-99)."""
    return _LinearTanhFn.apply(x, weight, bias, apply_tanh)


class _TpsGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, consts, h, w, npts):
        L = lib()
        Li, px, py, gx, gy = consts
        theta = theta.contiguous()
        b = theta.shape[0]
        grid = torch.empty((b, h, w, 2), dtype=torch.float32, device=theta.device)
        ws = workspace(theta.device, L.so_tps_ws_floats(b, h, w, npts) * 4)
        check(L.so_tps_grid_fwd(theta.data_ptr(), Li.data_ptr(), px.data_ptr(), py.data_ptr(), gx.data_ptr(), gy.data_ptr(),
                                grid.data_ptr(), b, h, w, npts, ws.data_ptr(), _stream()), "tps_fwd")
        ctx.consts = consts
        ctx.cfg = (b, h, w, npts)
        return grid

    @staticmethod
    def backward(ctx, dgrid):
        L = lib()
        Li, px, py, gx, gy = ctx.consts
        b, h, w, npts = ctx.cfg
        dgrid = dgrid.contiguous()
        dtheta = torch.empty((b, 2 * npts), dtype=torch.float32, device=dgrid.device)
        ws = workspace(dgrid.device, L.so_tps_ws_floats(b, h, w, npts) * 4)
        check(L.so_tps_grid_bwd(dgrid.data_ptr(), Li.data_ptr(), px.data_ptr(), py.data_ptr(), gx.data_ptr(), gy.data_ptr(),
                                dtheta.data_ptr(), b, h, w, npts, ws.data_ptr(), _stream()), "tps_bwd")
        return dtheta, None, None, None, None


def tps_grid(theta, consts, h, w, npts):
    return _TpsGridFn.apply(theta, consts, h, w, npts)


class _GridSampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, grid, border):
        _require_cuda(inp)
        inp = to_nchw(inp)
        grid = grid.contiguous()
        b, c, h, w = inp.shape
        ho, wo = grid.shape[1], grid.shape[2]
        out = torch.empty((b, c, ho, wo), dtype=torch.float32, device=inp.device)
        check(lib().so_grid_sample_fwd(inp.data_ptr(), grid.data_ptr(), out.data_ptr(), None, b, c, h, w, ho, wo, int(border), _stream()), "grid_sample_fwd")
        ctx.save_for_backward(inp, grid)
        ctx.border = border
        return out

    @staticmethod
    def backward(ctx, dout):
        inp, grid = ctx.saved_tensors
        b, c, h, w = inp.shape
        ho, wo = grid.shape[1], grid.shape[2]
        dout = to_nchw(dout)
        dgrid = torch.empty_like(grid) if ctx.needs_input_grad[1] else None
        din = None
        if ctx.needs_input_grad[0]:
            din = torch.empty_like(inp)
            check(lib().so_fill(din.data_ptr(), din.numel(), 0.0, _stream()), "fill")
        check(
            lib().so_grid_sample_bwd(inp.data_ptr(), grid.data_ptr(), dout.data_ptr(),
                                     dgrid.data_ptr() if dgrid is not None else None,
                                     din.data_ptr() if din is not None else None,
                                     b, c, h, w, ho, wo, int(ctx.border), _stream()),
            "grid_sample_bwd",
        )
        return din, dgrid, None


def grid_sample(inp, grid, padding_mode="zeros"):
    """F.grid_sample(inp, grid, mode='bilinear', padding_mode=..., align_corners=False)."""
    if padding_mode not in ("zeros", "border"):
        raise NotImplementedError(padding_mode)
    if inp.is_cuda and not inp.is_contiguous():
        inp = to_nchw(inp)
    return _GridSampleFn.apply(inp, grid, padding_mode == "border")


def grid_sample_taps(inp, grid, padding_mode="zeros"):
    """Forward only; also returns the int32 north-west tap indices (x0, y0) per output pixel."""
    inp = to_nchw(inp)
    grid = grid.contiguous()
    b, c, h, w = inp.shape
    ho, wo = grid.shape[1], grid.shape[2]
    out = torch.empty((b, c, ho, wo), dtype=torch.float32, device=inp.device)
    taps = torch.empty((b, ho, wo, 2), dtype=torch.int32, device=inp.device)
    check(lib().so_grid_sample_fwd(inp.data_ptr(), grid.data_ptr(), out.data_ptr(), taps.data_ptr(), b, c, h, w, ho, wo,
                                   int(padding_mode == "border"), _stream()), "grid_sample_fwd")
    return out, taps


class _Resample2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, flow):
        inp, flow = to_nchw(inp), to_nchw(flow)
        b, c, h, w = inp.shape
        out = torch.empty_like(inp)
        check(lib().so_resample2d_fwd(inp.data_ptr(), flow.data_ptr(), out.data_ptr(), b, c, h, w, _stream()), "resample2d_fwd")
        ctx.save_for_backward(inp, flow)
        return out

    @staticmethod
    def backward(ctx, dout):
        inp, flow = ctx.saved_tensors
        b, c, h, w = inp.shape
        dout = to_nchw(dout)
        din = dflow = None
        L = lib()
        if ctx.needs_input_grad[0]:
            din = torch.empty_like(inp)
        if ctx.needs_input_grad[1]:
            dflow = torch.empty_like(flow)
        ws = workspace(inp.device, L.so_resample2d_bwd_ws_floats(b, c, h, w) * 4, lane=7)
        check(
            L.so_resample2d_bwd(inp.data_ptr(), flow.data_ptr(), dout.data_ptr(),
                                din.data_ptr() if din is not None else None,
                                dflow.data_ptr() if dflow is not None else None, b, c, h, w, ws.data_ptr(), _stream()),
            "resample2d_bwd",
        )
        return din, dflow


def resample2d(inp, flow):
    return _Resample2dFn.apply(inp, flow)


# ------------------------------------------------------------------------------------------------
# losses / composition
# ------------------------------------------------------------------------------------------------
class _L1LossFn(torch.autograd.Function):
    """mean |a - b| * weight  (gradient flows to `a` only, as at every reference call site)."""

    @staticmethod
    def forward(ctx, a, b, weight):
        L = lib()
        _require_cuda(a)
        if a.shape != b.shape:
            raise ValueError("l1_loss: shape mismatch")
        rows_mode = a.dim() == 4 and not (a.is_contiguous() and b.is_contiguous())
        if rows_mode:
            a2, b2 = to_rows(a), to_rows(b)
            n, c, h, w = a2.shape
            lda, ldb, rows = _ld(a2), _ld(b2), n * h * w
        else:
            a2, b2 = a.contiguous(), b.contiguous()
            rows, c = 1, a2.numel()
            lda = ldb = c
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ws = workspace(a.device)
        scale = weight / a.numel()
        check(L.so_l1_loss_fwd(a2.data_ptr(), lda, b2.data_ptr(), ldb, rows, c, scale, out.data_ptr(), 0, ws.data_ptr(), _stream()), "l1_fwd")
        ctx.save_for_backward(a2, b2)
        ctx.cfg = (lda, ldb, rows, c, scale, tuple(a.shape), rows_mode)
        return out

    @staticmethod
    def backward(ctx, gout):
        a2, b2 = ctx.saved_tensors
        lda, ldb, rows, c, scale, shape, rows_mode = ctx.cfg
        gout = gout.contiguous()
        if rows_mode:
            n, ch, h, w = shape
            da = nhwc_empty(n, h, w, ch, a2.device)
        else:
            da = torch.empty(shape, dtype=torch.float32, device=a2.device)
        check(lib().so_l1_loss_bwd(a2.data_ptr(), lda, b2.data_ptr(), ldb, gout.data_ptr(), scale, da.data_ptr(), c, rows, c, 0, 0, _stream()), "l1_bwd")
        return da, None, None


def l1_loss(a, b, weight=1.0):
    return _L1LossFn.apply(a, b, float(weight))


class _ScalarSumFn(torch.autograd.Function):
    """((a + b) + c) + d of 0-dim loss terms in one launch - the same left-to-right additions as `a + b + c + d`, each of
    which is a kernel of its own inside the captured step; every term's gradient is the incoming one."""

    @staticmethod
    def forward(ctx, *terms):
        out = torch.empty((), dtype=torch.float32, device=terms[0].device)
        p = [t.data_ptr() for t in terms] + [None] * (4 - len(terms))
        check(lib().so_scalar_sum(p[0], p[1], p[2], p[3], out.data_ptr(), _stream()), "scalar_sum")
        ctx.n = len(terms)
        return out

    @staticmethod
    def backward(ctx, gout):
        return (gout,) * ctx.n


_ZERO = {}


def zero_scalar(device):
    """A constant 0-dim zero (the n_frames_total = 1 placeholders of the *_prev terms and of the flow-mask penalty)."""
    key = (device.type, device.index)
    z = _ZERO.get(key)
    if z is None:
        z = _ZERO[key] = torch.zeros((), dtype=torch.float32, device=device)
    return z


_ONE = {}


def one_scalar(device):
    """A constant 0-dim one: the root gradient of `loss.backward(...)` (autograd otherwise launches a fill for it every step)."""
    key = (device.type, device.index)
    o = _ONE.get(key)
    if o is None:
        o = _ONE[key] = torch.ones((), dtype=torch.float32, device=device)
    return o


def backward(loss):
    """loss.backward() with the cached unit root gradient (one ATen fill launch less per backward pass)."""
    if loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32:
        loss.backward(one_scalar(loss.device))
    else:
        loss.backward()


def scalar_sum(*terms):
    """Sum of 2..4 scalar loss terms, added left to right (bit-identical to chained `+`)."""
    if not (2 <= len(terms) <= 4) or not all(t.is_cuda and t.dim() == 0 and t.dtype == torch.float32 for t in terms):
        out = terms[0]
        for t in terms[1:]:
            out = out + t
        return out
    return _ScalarSumFn.apply(*terms)


class _TryonComposeFn(torch.autograd.Function):
    """tanh / sigmoid / blend of UnetMaskModel.forward for one frame (unet_mask_model.py:84-86,126-129)."""

    @staticmethod
    def forward(ctx, o, cloth):
        L = lib()
        ctx.set_materialize_grads(False)   # p_rendered carries no loss term (unet_mask_model.py:165-188): no zero fill for it
        o, cloth = to_rows(o), to_rows(cloth)
        n, c, h, w = o.shape
        dev = o.device
        rendered = nhwc_empty(n, h, w, 3, dev, ld=4)
        mask = nhwc_empty(n, h, w, 1, dev)
        tryon = nhwc_empty(n, h, w, 3, dev, ld=4)
        check(
            L.so_tryon_compose_fwd(o.data_ptr(), _ld(o), cloth.data_ptr(), _ld(cloth), rendered.data_ptr(), 4,
                                   mask.data_ptr(), 1, tryon.data_ptr(), 4, 4, n * h * w, _stream()),
            "compose_fwd",
        )
        ctx.save_for_backward(rendered, mask, cloth)
        ctx.cfg = (n, c, h, w)
        return rendered, mask, tryon

    @staticmethod
    def backward(ctx, d_rendered, d_mask, d_tryon):
        rendered, mask, cloth = ctx.saved_tensors
        n, c, h, w = ctx.cfg
        dev = rendered.device
        d_o = nhwc_empty(n, h, w, c, dev)
        if c > 4:
            check(lib().so_fill(d_o.data_ptr(), d_o.numel(), 0.0, _stream()), "fill")

        def prep(t):
            if t is None:
                return None, 0
            t = to_rows(t)
            return t, _ld(t)

        dr, lddr = prep(d_rendered)
        dm, lddm = prep(d_mask)
        dt, lddt = prep(d_tryon)
        check(
            lib().so_tryon_compose_bwd(
                rendered.data_ptr(), 4, mask.data_ptr(), 1, cloth.data_ptr(), _ld(cloth),
                dt.data_ptr() if dt is not None else None, lddt,
                dr.data_ptr() if dr is not None else None, lddr,
                dm.data_ptr() if dm is not None else None, lddm,
                d_o.data_ptr(), c, n * h * w, _stream(),
            ),
            "compose_bwd",
        )
        return d_o, None


def tryon_compose(o, cloth):
    return _TryonComposeFn.apply(o, cloth)


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    """Fused Adam on flat fp32 slabs (torch.optim.Adam semantics, base_model.py:165-168)."""
    check(lib().so_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                             int(step), grad_scale, _stream()), "adam_step")


def fill_(t, value=0.0):
    check(lib().so_fill(t.data_ptr(), t.numel(), float(value), _stream()), "fill")
    return t


class _BlendFn(torch.autograd.Function):
    """(1 - m) * a + m * b with a one-channel mask m (multi-frame path, unet_mask_model.py:118-129)."""

    @staticmethod
    def forward(ctx, a, b, m):
        a, b, m = to_rows(a), to_rows(b), to_rows(m)
        n, c, h, w = a.shape
        out = nhwc_empty(n, h, w, c, a.device)
        check(lib().so_blend_fwd(a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), m.data_ptr(), _ld(m), out.data_ptr(), c,
                                 n * h * w, c, _stream()), "blend_fwd")
        ctx.save_for_backward(a, b, m)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, m = ctx.saved_tensors
        n, c, h, w = a.shape
        g = to_rows(g)
        dev = a.device
        da = nhwc_empty(n, h, w, c, dev) if ctx.needs_input_grad[0] else None
        db = nhwc_empty(n, h, w, c, dev) if ctx.needs_input_grad[1] else None
        dm = nhwc_empty(n, h, w, 1, dev) if ctx.needs_input_grad[2] else None
        p = lambda t: t.data_ptr() if t is not None else None
        check(lib().so_blend_bwd(a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), m.data_ptr(), _ld(m), g.data_ptr(), _ld(g),
                                 p(da), c, p(db), c, p(dm), 1, n * h * w, c, _stream()), "blend_bwd")
        return da, db, dm


def blend(a, b, m):
    return _BlendFn.apply(a, b, m)


class _SumFn(torch.autograd.Function):
    """tensor.sum() (flow-mask penalty, unet_mask_model.py:186-188) as <x, 1>."""

    @staticmethod
    def forward(ctx, x):
        x2 = to_rows(x) if x.dim() == 4 else x.contiguous()
        ones = torch.empty(x2.numel(), dtype=torch.float32, device=x.device)
        fill_(ones, 1.0)
        xc = _copy_rows(x2) if x.dim() == 4 and _ld(x2) != x2.shape[1] else x2
        out = torch.empty((), dtype=torch.float32, device=x.device)
        ws = workspace(x.device)
        check(lib().so_dot(xc.data_ptr(), xc.numel(), ones.data_ptr(), xc.numel(), 1, xc.numel(), 1.0, out.data_ptr(),
                           0, ws.data_ptr(), _stream()), "dot")
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        shape = ctx.shape
        if len(shape) == 4:
            out = nhwc_empty(shape[0], shape[2], shape[3], shape[1], g.device)
        else:
            out = torch.empty(shape, dtype=torch.float32, device=g.device)
        n = out.numel()
        ones = fill_(torch.empty(n, dtype=torch.float32, device=g.device), 1.0)
        zeros = fill_(torch.empty(n, dtype=torch.float32, device=g.device), 0.0)
        g = g.contiguous()
        # out = g * 1 + 0 (g is a device scalar)
        check(lib().so_scale_add(ones.data_ptr(), n, g.data_ptr(), zeros.data_ptr(), n, out.data_ptr(), n, 1, n,
                                 _stream()), "scale_add")
        return out


def tensor_sum(x):
    return _SumFn.apply(x)



_SB16_W_CACHE = {}


def _sb16_eligible(ci, co):
    return ci % 32 == 0 and co % 64 == 0


def _sb16_planes(t2d_ptr, ld, c, rows, device):
    """fp32 rows -> (hi, mid) bf16 planes [rows][c]."""
    hi = torch.empty((rows, c), dtype=torch.bfloat16, device=device)
    mid = torch.empty((rows, c), dtype=torch.bfloat16, device=device)
    check(lib().so_sb16_split(t2d_ptr, ld, c, hi.data_ptr(), mid.data_ptr(), rows, _stream()), "sb16_split")
    return hi, mid


def _sb16_weights(wk, owner, transpose):
    """(hi, mid) planes of the frozen OHWI weights `wk` (Ko, 3, 3, C): [Ko][9][C], or for the input gradient [C][9][Ko] with
    flipped taps.  Cached per parameter object and version like the transposed fp32 copies."""
    ref, version = owner
    p = ref()
    version = (version, p.data_ptr()) if p is not None else version
    key = (id(p), bool(transpose))
    hit = _SB16_W_CACHE.get(key) if p is not None else None
    if hit is not None and hit[0]() is p and hit[1] == version:
        return _serve_cached(hit[2], hit[3])
    if torch.cuda.is_current_stream_capturing():
        p = None       # see _ihwo: no entry is created under capture
    ko, _, _, c = wk.shape
    hi = torch.empty(ko * 9 * c, dtype=torch.bfloat16, device=wk.device)
    mid = torch.empty_like(hi)
    check(lib().so_sb16_prep_weights(wk.data_ptr(), ko, c, c, int(transpose), hi.data_ptr(), mid.data_ptr(), _stream()), "sb16_prep")
    if p is not None:
        for k in [k for k, v in _SB16_W_CACHE.items() if v[0]() is None]:
            del _SB16_W_CACHE[k]
        _SB16_W_CACHE[key] = (ref, version, hi, mid)
    return hi, mid

_WINO_W_CACHE = {}


def invalidate_weight_caches():
    """Drop every derived-weight cache (Winograd-domain / transposed / split-bf16 / stacked q-k-v copies of parameters that no
    gradient is asked of).  The caches follow `Tensor._version` and `data_ptr()`; a write that changes neither - `p.data.copy_`,
    `dist.broadcast(p.data)`, the reference-style `m.weight.data.normal_()` - needs this call (trainer.broadcast_parameters
    and BaseModel.load_state_dict make it).  A graphed step captured before the call reads pinned copies of the old weights and
    must be rebuilt (its replay raises)."""
    for c in (_WINO_W_CACHE, _IHWO_CACHE, _SB16_W_CACHE, _QKV_CACHE):
        c.clear()
    _CACHE_GENERATION[0] += 1   # steps captured earlier may hold pinned copies of the OLD weights: graphs.py refuses to replay them


def _wino_mode(ci, co, n, h, w):
    """Which kernel computes a 3x3 / s1 / p1 convolution with ci -> co channels on n x h x w pixels: "direct", "fused" or
    "nonfused".  The auto rule is the measured crossover on MI355X (tools/wino_bench.py, profiles/r03_wino_bench.csv)."""
    if WINOGRAD in ("0", "off", "direct") or ci % 4 or co % 4 or min(ci, co) < 32:
        return "direct"
    if WINOGRAD in ("fused", "nonfused"):
        return WINOGRAD
    tw, th = (w + 1) // 2, (h + 1) // 2
    blocks = n * ((tw + 7) // 8) * ((th + 3) // 4) * ((co + 31) // 32)   # fused kernel: one block per (8x4-tile patch, 32 ko)
    # measured (profiles/r03_wino_bench.csv): the fused kernel wins while the transformed operands of the non-fused form
    # (16 / 4 x the activation size, written and read once each) outweigh its better GEMM: up to 128 channels from 64x48x4
    # pixels, up to 256 channels from 64x48x8 pixels; the deep layers (256..512 channels at <= 32x24) go to the 16-GEMM batch
    px, cmin = n * h * w, min(ci, co)
    if WINOGRAD == "auto" and WINOGRAD_F44 and cmin >= 256 and px <= 24576 and h % 4 == 0 and w % 4 == 0:
        return "nonfused4"   # F(4x4, 3x3): the deep layers (>= 256 channels at <= 64x48): 2.25 instead of 4 multiplications
    if blocks >= 768 and ((cmin <= 128 and px >= 12288) or (cmin <= 256 and px >= 24576)):
        return "fused"
    return "nonfused" if cmin >= 128 else ("fused" if blocks >= 768 else "direct")


def _wino_weights(wk, owner, transpose, fused=False, ko_pad=None, f44=False):
    """Winograd-domain weights of the OHWI tensor `wk` (Ko, 3, 3, C): U[16][Ko][C], or for the input gradient U'[16][C][Ko]
    (flipped taps); fused=True: the fused kernel's [K/8][16][N][8] order.  `owner` = (weakref to the parameter, its version): frozen weights are transformed once and cached per
    parameter object and version like the transposed copies; owner None transforms on every call (trainable weights)."""
    kw, _, _, c = wk.shape
    ko = kw if ko_pad is None else ko_pad   # output channels zero-padded to a multiple of 4 (rows >= kw are zero)
    L = lib()
    key = None
    numel = L.so_wino_fused_weight_floats(ko, c, int(transpose)) if fused else (36 if f44 else 16) * ko * c
    if owner is not None:
        ref, version = owner
        p = ref()
        owner = (ref, (version, p.data_ptr()) if p is not None else version)
        key = (id(p), bool(transpose), bool(fused), bool(f44))
        hit = _WINO_W_CACHE.get(key) if p is not None else None
        if hit is not None and hit[0]() is p and hit[1] == owner[1] and hit[2].numel() == numel:
            return _serve_cached(hit[2])
        if torch.cuda.is_current_stream_capturing():
            key = None   # see _ihwo: no entry is created under capture
    u = torch.empty(numel, dtype=torch.float32, device=wk.device)
    fn = L.so_wino_fused_weights if fused else (L.so_wino4_weights if f44 else L.so_wino_weights)
    check(fn(wk.data_ptr(), u.data_ptr(), ko, kw, c, int(transpose), _stream()), "wino_weights")
    if key is not None and owner[0]() is not None:
        for k in [k for k, v in _WINO_W_CACHE.items() if v[0]() is None]:
            del _WINO_W_CACHE[k]
        _WINO_W_CACHE[key] = (owner[0], owner[1], u)
    return u


def wino_conv3x3(x_ptr, ldx, u, bias, gate_ptr, y_ptr, ldy, n, h, w, ci, co, act, device, lane=4, fused=False, act_param=0.0,
                 f44=False):
    """y = gate(act(conv3x3(x) + bias)) through F(2x2,3x3); u = _wino_weights(...) with N = co rows of K = ci columns."""
    L = lib()
    if fused:
        check(L.so_wino_fused_conv3x3(x_ptr, ldx, u.data_ptr(), bias.data_ptr() if bias is not None else None,
                                      bias.numel() if bias is not None else 0, gate_ptr, y_ptr, ldy, n, h, w, ci, co, act,
                                      float(act_param), _stream()), "wino_fused_conv3x3")
        return
    need = (L.so_wino4_ws_floats if f44 else L.so_wino_ws_floats)(n, h, w, ci, co) * 4
    wws = workspace(device, need, lane=lane)   # transformed operands: their own slab (split-K slabs live in lane 0)
    ws = workspace(device)
    check((L.so_wino4_conv3x3 if f44 else L.so_wino_conv3x3)(x_ptr, ldx, u.data_ptr(), bias.data_ptr() if bias is not None else None,
                            bias.numel() if bias is not None else 0, gate_ptr, y_ptr, ldy, n, h, w, ci, co, act, float(act_param),
                            wws.data_ptr(), wws.numel() * 4, ws.data_ptr(), ws.numel() * 4, _stream()), "wino_conv3x3")


# ------------------------------------------------------------------------------------------------
# VGG19 perceptual loss as ONE autograd node (models/networks/loss.py:106-122 + vgg.py:6-36)
# ------------------------------------------------------------------------------------------------
class _VggLossFn(torch.autograd.Function):
    """sum_i w_i * mean|f_i(x) - f_i(y)| over the five relu taps, VGG weights frozen.

    The prediction x and the target y run through the stack as ONE batch (2B images: deeper layers get a
    taller GEMM), Conv3x3+ReLU is a single MFMA kernel, and the backward pass walks the x half only:
    L1 sign gradient injected at each tap, ReLU mask from the saved activations, dgrad (no wgrad: frozen).
    `cfg` is a tuple of ("C", tap_weight or None) / ("M",) items; `params` = (w0, b0, w1, b1, ...).

    With `nfeat` > 0 the first `nfeat` entries of `rest` are the target's tap features computed beforehand
    (vgg_target_features: they depend on the batch only, so a pipeline can produce them ahead of time on another
    stream); then only x runs through the stack and y is ignored."""

    @staticmethod
    def forward(ctx, x, y, cfg, nfeat, *rest):
        L = lib()
        yfeats, params = rest[:nfeat], rest[nfeat:]
        xr = to_rows(x)
        b, c, h, w = xr.shape
        dev = xr.device
        cp = (c + 3) // 4 * 4
        nb = b if nfeat else 2 * b
        inp = nhwc_empty(nb, h, w, cp, dev)
        half = b * h * w * cp * 4
        check(L.so_copy2d(xr.data_ptr(), _ld(xr), c, inp.data_ptr(), cp, cp, b * h * w, 0, _stream()), "copy2d")
        if not nfeat:
            yr = to_rows(y.detach())
            check(L.so_copy2d(yr.data_ptr(), _ld(yr), c, inp.data_ptr() + half, cp, cp, b * h * w, 0, _stream()), "copy2d")
        ws = workspace(dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        fill_(loss, 0.0)
        cur, saved, meta, pi, ti = inp, [], [], 0, 0
        relu_in = None  # index in `saved` of the ReLU output that is the current tensor (None: image / pooled map)
        split = VGG_SPLIT_BF16
        planes = None   # (hi, mid) bf16 planes of `cur` when the producing convolution emitted them
        skip_pool = False
        for ii, item in enumerate(cfg):
            n2, ci, hh, ww = cur.shape
            if item[0] == "M" and skip_pool:   # written by the preceding convolution's epilogue (below)
                skip_pool = False
                continue
            if item[0] == "M":
                planes = None
                out = nhwc_empty(n2, hh // 2, ww // 2, ci, dev)
                check(L.so_maxpool2_fwd(cur.data_ptr(), ci, out.data_ptr(), ci, n2, hh, ww, ci, _stream()), "maxpool2_fwd")
                saved.append(cur)
                meta.append(("M", len(saved) - 1, relu_in is not None))
                relu_in = None
            else:
                weight, bias = params[pi], params[pi + 1]
                pi += 2
                co = weight.shape[0]
                wk = _ohwi(weight, cpad=ci)
                out = nhwc_empty(n2, hh, ww, co, dev)
                wowner = (weakref.ref(weight), weight._version)
                if split and _sb16_eligible(ci, co):
                    # split-bf16 path: input planes come from the previous convolution's epilogue when there was one
                    if planes is None:
                        planes = _sb16_planes(cur.data_ptr(), ci, ci, n2 * hh * ww, dev)
                    wh, wm = _sb16_weights(wk, wowner, transpose=False)
                    nxt = cfg[ii + 1] if ii + 1 < len(cfg) else None
                    emit = nxt is not None and nxt[0] == "C" and _sb16_eligible(co, params[pi].shape[0])
                    oh = torch.empty((n2 * hh * ww, co), dtype=torch.bfloat16, device=dev) if emit else None
                    om = torch.empty_like(oh) if emit else None
                    check(L.so_sb16_conv3x3(planes[0].data_ptr(), planes[1].data_ptr(), wh.data_ptr(), wm.data_ptr(), bias.data_ptr(),
                                            None, out.data_ptr(), co, oh.data_ptr() if emit else None, om.data_ptr() if emit else None,
                                            n2, hh, ww, ci, co, 1, _stream()), "sb16_conv3x3")
                    planes = (oh, om) if emit else None
                elif _wino_mode(ci, co, n2, hh, ww) != "direct":
                    wm = _wino_mode(ci, co, n2, hh, ww)
                    fz, f4 = wm == "fused", wm == "nonfused4"
                    nxt = cfg[ii + 1] if ii + 1 < len(cfg) else None
                    if fz and item[1] is None and nxt is not None and nxt[0] == "M" and hh % 2 == 0 and ww % 2 == 0:
                        # conv -> ReLU -> MaxPool2d(2, 2) (vgg.py: conv1_2, conv2_2): an F(2x2) output tile IS a pooling window,
                        # so the epilogue writes the pooled map too, and the un-pooled activations only of the prediction
                        # half (the backward pass gates on them; nobody reads the target's)
                        out = nhwc_empty(b, hh, ww, co, dev)
                        pooled = nhwc_empty(n2, hh // 2, ww // 2, co, dev)
                        u = _wino_weights(wk, wowner, False, True)
                        check(L.so_wino_fused_conv3x3_pool(cur.data_ptr(), ci, u.data_ptr(), bias.data_ptr(), bias.numel(), out.data_ptr(),
                                                           co, b, pooled.data_ptr(), co, n2, hh, ww, ci, co, ACT_RELU, 0.0, _stream()),
                              "wino_fused_conv3x3_pool")
                        saved.extend([out, wk])
                        meta.append(("C", len(saved) - 2, len(saved) - 1, None, ci, (weakref.ref(weight), weight._version), relu_in, None))
                        saved.append(out)
                        meta.append(("M", len(saved) - 1, True))
                        relu_in, planes, cur, skip_pool = None, None, pooled, True
                        continue
                    wino_conv3x3(cur.data_ptr(), ci, _wino_weights(wk, wowner, False, fz, f44=f4), bias, None, out.data_ptr(), co, n2,
                                 hh, ww, ci, co, ACT_RELU, dev, fused=fz, f44=f4)
                    planes = None
                else:
                    check(L.so_conv2d_fprop(cur.data_ptr(), ci, wk.data_ptr(), bias.data_ptr(), out.data_ptr(), co, n2, hh, ww, ci, co,
                                            3, 3, 1, 1, ACT_RELU, 0.0, ws.data_ptr(), ws.numel() * 4, _stream()), "conv2d_fprop")
                    planes = None
                tap_w = item[1]
                ytap = None  # index in `saved` of the precomputed target feature of this tap
                if tap_w is not None:
                    rows = b * hh * ww
                    yp = out.data_ptr() + rows * co * 4
                    if nfeat:
                        yf = yfeats[ti]
                        ti += 1
                        if tuple(yf.shape) != (b, co, hh, ww) or _ld(yf) != co:
                            raise RuntimeError("precomputed VGG target feature has the wrong shape / layout")
                        yp = yf.data_ptr()
                        saved.append(yf)
                        ytap = len(saved) - 1
                    check(L.so_l1_loss_fwd(out.data_ptr(), co, yp, co, rows, co, tap_w / (rows * co),
                                           loss.data_ptr(), 1, ws.data_ptr(), _stream()), "l1_fwd")
                saved.extend([out, wk])
                meta.append(("C", len(saved) - 2, len(saved) - 1, tap_w, ci, (weakref.ref(weight), weight._version), relu_in, ytap))
                relu_in = len(saved) - 2
            cur = out
        ctx.save_for_backward(*saved)
        ctx.meta = (meta, b, c, cp)
        ctx.split = split
        return loss

    @staticmethod
    def backward(ctx, gout):
        L = lib()
        saved = ctx.saved_tensors
        meta, b, c, cp = ctx.meta
        dev = gout.device
        gout = gout.contiguous()
        ws = workspace(dev)
        # `g` is always the gradient IN FRONT of the current layer's ReLU: every producer applies that ReLU's mask itself
        # (dgrad epilogue gate, max-pool backward gate, L1 backward gate), so no separate mask pass runs.
        g = None
        for m in reversed(meta):
            if m[0] == "M":
                xin = saved[m[1]]
                _, ci, hh, ww = xin.shape
                dx = nhwc_empty(b, hh, ww, ci, dev)
                check(L.so_maxpool2_bwd(xin.data_ptr(), ci, g.data_ptr(), ci, dx.data_ptr(), ci, b, hh, ww, ci, int(m[2]), _stream()), "maxpool2_bwd")
                g = dx
                continue
            _, oi, wi, tap_w, ci, wkey, relu_in, ytap = m
            out, wk = saved[oi], saved[wi]
            _, co, hh, ww = out.shape
            rows = b * hh * ww
            if tap_w is not None:
                acc = 1
                if g is None:
                    g, acc = nhwc_empty(b, hh, ww, co, dev), 0
                yp = saved[ytap].data_ptr() if ytap is not None else out.data_ptr() + rows * co * 4
                check(L.so_l1_loss_bwd(out.data_ptr(), co, yp, co, gout.data_ptr(), tap_w / (rows * co),
                                       g.data_ptr(), co, rows, co, acc, 1, _stream()), "l1_bwd")
            dx = nhwc_empty(b, hh, ww, ci, dev)
            gate = saved[relu_in].data_ptr() if relu_in is not None else None  # x half = first b images of the 2b batch
            if ctx.split and _sb16_eligible(co, ci):
                # input gradient = the same split-bf16 convolution on flipped + transposed weight planes ([ci][9][co])
                gh, gm = _sb16_planes(g.data_ptr(), co, co, rows, dev)
                wh, wm = _sb16_weights(wk, wkey, transpose=True)
                check(L.so_sb16_conv3x3(gh.data_ptr(), gm.data_ptr(), wh.data_ptr(), wm.data_ptr(), None, gate, dx.data_ptr(), ci,
                                        None, None, b, hh, ww, co, ci, 0, _stream()), "sb16_dgrad")
            elif _wino_mode(co, ci, b, hh, ww) != "direct":
                # input gradient = the same Winograd convolution on flipped taps with C and Ko swapped (U'[16][ci][co])
                wm = _wino_mode(co, ci, b, hh, ww)
                fz, f4 = wm == "fused", wm == "nonfused4"
                wino_conv3x3(g.data_ptr(), co, _wino_weights(wk, wkey, True, fz, f44=f4), None, gate, dx.data_ptr(), ci, b, hh, ww, co,
                             ci, ACT_NONE, dev, fused=fz, f44=f4)
            else:
                wt = _ihwo(wk, owner=wkey)  # frozen weights: transposed once, reused every step
                check(L.so_conv2d_dgrad_t_gated(g.data_ptr(), co, wt.data_ptr(), dx.data_ptr(), ci, gate, b, hh, ww, ci, co, 3, 3, 1, 1,
                                                ws.data_ptr(), ws.numel() * 4, _stream()), "conv2d_dgrad_t")
            g = dx
        dxr = g if cp == c else g[:, :c]
        return (dxr, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


def vgg_perceptual_loss(x, y, cfg, params, y_features=None):
    feats = tuple(y_features) if y_features is not None else ()
    return _VggLossFn.apply(x, y if not feats else None, cfg, len(feats), *feats, *params)


@torch.no_grad()
def vgg_target_features(y, cfg, params):
    """The tap features (relu1_1 ... relu5_1 outputs, NHWC) of the target image for vgg_perceptual_loss(y_features=...)."""
    L = lib()
    yr = to_rows(y.detach())
    b, c, h, w = yr.shape
    dev = yr.device
    cp = (c + 3) // 4 * 4
    cur = nhwc_empty(b, h, w, cp, dev)
    check(L.so_copy2d(yr.data_ptr(), _ld(yr), c, cur.data_ptr(), cp, cp, b * h * w, 0, _stream()), "copy2d")
    ws = workspace(dev)
    feats, pi = [], 0
    for item in cfg:
        n2, ci, hh, ww = cur.shape
        if item[0] == "M":
            out_pool = nhwc_empty(n2, hh // 2, ww // 2, ci, dev)
            check(L.so_maxpool2_fwd(cur.data_ptr(), ci, out_pool.data_ptr(), ci, n2, hh, ww, ci, _stream()), "maxpool2_fwd")
        else:
            weight, bias = params[pi], params[pi + 1]
            pi += 2
            co = weight.shape[0]
            wk = _ohwi(weight, cpad=ci)
            o_ = nhwc_empty(n2, hh, ww, co, dev)
            mode = _wino_mode(ci, co, n2, hh, ww)
            if mode != "direct":
                wino_conv3x3(cur.data_ptr(), ci, _wino_weights(wk, (weakref.ref(weight), weight._version), False, mode == "fused",
                                                               f44=mode == "nonfused4"),
                             bias, None, o_.data_ptr(), co, n2, hh, ww, ci, co, ACT_RELU, dev, fused=mode == "fused",
                             f44=mode == "nonfused4")
            else:
                check(L.so_conv2d_fprop(cur.data_ptr(), ci, wk.data_ptr(), bias.data_ptr(), o_.data_ptr(), co, n2, hh, ww, ci, co,
                                        3, 3, 1, 1, ACT_RELU, 0.0, ws.data_ptr(), ws.numel() * 4, _stream()), "conv2d_fprop")
            if item[1] is not None:
                feats.append(o_)
            cur = o_
            continue
        cur = out_pool
    return feats


# ------------------------------------------------------------------------------------------------
# scratch lanes follow an operator into its backward pass
# ------------------------------------------------------------------------------------------------
def _lane_aware(cls):
    """An operator issued under `workspace_lane(b)` (a branch running concurrently on another stream) must use the same scratch
    lane in its BACKWARD pass - autograd runs it later, outside the `with` block, on that branch's stream, concurrently with the
    other branch's backward nodes.  forward() records the lane base in ctx, backward() re-enters it."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx._so_lane_base = _LANE_BASE[0]
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        prev = _LANE_BASE[0]
        _LANE_BASE[0] = getattr(ctx, "_so_lane_base", prev)
        try:
            return bwd(ctx, *grads)
        finally:
            _LANE_BASE[0] = prev

    forward.__doc__, backward.__doc__ = fwd.__doc__, bwd.__doc__
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


def make_functions_lane_aware(namespace):
    for obj in list(namespace.values()):
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function \
                and not getattr(obj, "_so_lane_aware", False) and obj.__module__ == namespace.get("__name__"):
            _lane_aware(obj)
            obj._so_lane_aware = True


make_functions_lane_aware(globals())

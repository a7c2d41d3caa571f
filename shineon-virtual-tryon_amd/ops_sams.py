"""Autograd operators that only the SAMS-GAN path needs, over csrc/sams.hip.  Same contract as
ops.py: torch supplies device memory, the current stream and the autograd tape; the arithmetic is in libshineon_hip.so.
"""
import torch

from ._lib import check, lib
from .ops import (ACT_CODES, ACT_NONE, _direct_grad_ok, _ld, _require_cuda, _stream, grad_ready, nhwc_empty, to_rows, workspace)

GAN_MODES = {"original": 0, "ls": 1, "w": 2, "hinge": 3}


def _ohwi_dense(t):
    """(O, I, R, S) tensor -> the same values as a dense OHWI block (no copy when it already is one)."""
    v = t.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


# ------------------------------------------------------------------------------------------------
# nearest-neighbour resize
# ------------------------------------------------------------------------------------------------
class _ResizeNearestFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ho, wo, sh, sw):
        x = to_rows(x)
        n, c, h, w = x.shape
        y = nhwc_empty(n, ho, wo, c, x.device)
        check(lib().so_resize_nearest_fwd(x.data_ptr(), _ld(x), y.data_ptr(), c, n, h, w, ho, wo, c, sh, sw, _stream()),
              "resize_nearest_fwd")
        ctx.cfg = (n, c, h, w, ho, wo, sh, sw)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w, ho, wo, sh, sw = ctx.cfg
        dy = to_rows(dy)
        dx = nhwc_empty(n, h, w, c, dy.device)
        check(lib().so_resize_nearest_bwd(dy.data_ptr(), _ld(dy), dx.data_ptr(), c, n, h, w, ho, wo, c, sh, sw, _stream()),
              "resize_nearest_bwd")
        return dx, None, None, None, None


def resize_nearest(x, size=None, scale_factor=None):
    """F.interpolate(x, size=..., mode="nearest") or nn.Upsample(scale_factor=...) (default mode "nearest").
    The source-index scale follows ATen: in / out for `size`, 1 / scale_factor for `scale_factor`."""
    _require_cuda(x)
    h, w = x.shape[2:]
    if (size is None) == (scale_factor is None):
        raise ValueError("exactly one of size / scale_factor")
    if size is not None:
        ho, wo = int(size[0]), int(size[1])
        sh, sw = h / ho, w / wo
    else:
        ho, wo = int(h * scale_factor), int(w * scale_factor)  # floor(in * scale), torch's output-size rule
        sh = sw = 1.0 / scale_factor
    if ho == h and wo == w:
        return x
    return _ResizeNearestFn.apply(x, ho, wo, float(sh), float(sw))


# ------------------------------------------------------------------------------------------------
# SPADE modulation
# ------------------------------------------------------------------------------------------------
class _SpadeFn(torch.autograd.Function):
    """y = act(n * (1 + gamma) + beta) with gamma | beta the two channel halves of ONE conv output `gb`."""

    @staticmethod
    def forward(ctx, nrm, gb, act, param):
        nrm, gb = to_rows(nrm), to_rows(gb)
        n, c, h, w = nrm.shape
        y = nhwc_empty(n, h, w, c, nrm.device)
        check(lib().so_spade_fwd(nrm.data_ptr(), _ld(nrm), gb.data_ptr(), _ld(gb), gb.data_ptr() + 4 * c, _ld(gb),
                                 y.data_ptr(), c, n * h * w, c, act, param, _stream()), "spade_fwd")
        ctx.save_for_backward(nrm, gb)
        ctx.cfg = (act, param)
        return y

    @staticmethod
    def backward(ctx, dy):
        nrm, gb = ctx.saved_tensors
        act, param = ctx.cfg
        dy = to_rows(dy)
        n, c, h, w = nrm.shape
        dgb = nhwc_empty(n, h, w, 2 * c, nrm.device)
        dn = nhwc_empty(n, h, w, c, nrm.device)
        L = lib()
        rows = n * h * w
        nb = L.so_spade_bwd_colsum_blocks(rows, c) if _ld(nrm) % 4 == 0 and _ld(gb) % 4 == 0 and _ld(dy) % 4 == 0 else 0
        part = torch.empty((nb, 2 * c), dtype=torch.float32, device=nrm.device) if nb else None
        check(L.so_spade_bwd(nrm.data_ptr(), _ld(nrm), gb.data_ptr(), _ld(gb), gb.data_ptr() + 4 * c, _ld(gb),
                             dy.data_ptr(), _ld(dy), dn.data_ptr(), c, dgb.data_ptr(), 2 * c, dgb.data_ptr() + 4 * c, 2 * c,
                             rows, c, act, param, part.data_ptr() if nb else None, _stream()), "spade_bwd")
        if nb:
            # column sums of dgamma | dbeta = the bias gradient of the convolution that produced them: merged from the
            # per-block partials (<= 1024 rows) instead of a second pass over the full gradient
            db = torch.empty(2 * c, dtype=torch.float32, device=nrm.device)
            ws = workspace(nrm.device, L.so_colsum_ws_floats(nb, 2 * c) * 4, lane=6)
            check(L.so_colsum(part.data_ptr(), 2 * c, nb, 2 * c, db.data_ptr(), 0, ws.data_ptr(), _stream()), "colsum")
            dgb._so_bias_grad = db
        return dn, dgb, None, None


def spade_modulate(normalized, gamma_beta, act=ACT_NONE, param=0.0):
    """SPADE.forward's last line (sams/spade.py:89), optionally fused with the activation that follows it in
    AnySpadeResBlock.forward (spade.py:168-169).  gamma_beta: (N, 2C, H, W), the output of the mlp_gamma and mlp_beta
    convolutions evaluated as one convolution with stacked weights."""
    _require_cuda(normalized)
    if gamma_beta.shape[1] != 2 * normalized.shape[1]:
        raise ValueError("gamma_beta must hold 2 * C channels")
    return _SpadeFn.apply(normalized, gamma_beta, ACT_CODES[act] if not isinstance(act, int) else act, float(param))


class _SpadeNormFn(torch.autograd.Function):
    """y = act(norm(x) * (1 + gamma) + beta) with the parameter-free batch / instance norm of SPADE folded into the
    modulation pass: statistics (so_norm_fwd without an output), then ONE pass that normalises, modulates and activates.
    The normalised tensor is never written; the backward pass recomputes it from x and the saved statistics."""

    @staticmethod
    def forward(ctx, x, gb, running_mean, running_var, instance, momentum, eps, act, param):
        L = lib()
        x, gb = to_rows(x), to_rows(gb)
        n, c, h, w = x.shape
        G, R = (n, h * w) if instance else (1, n * h * w)
        mean = torch.empty((G, c), dtype=torch.float32, device=x.device)
        rstd = torch.empty((G, c), dtype=torch.float32, device=x.device)
        ws = workspace(x.device, L.so_norm_ws_floats(G, R, c) * 4)
        check(L.so_norm_fwd(x.data_ptr(), _ld(x), None, 0, G, R, c, eps, None, None, mean.data_ptr(), rstd.data_ptr(),
                            running_mean.data_ptr() if running_mean is not None else None,
                            running_var.data_ptr() if running_var is not None else None, momentum, ws.data_ptr(), _stream()),
              "norm_stats")
        y = nhwc_empty(n, h, w, c, x.device)
        check(L.so_spade_norm_fwd(x.data_ptr(), _ld(x), mean.data_ptr(), rstd.data_ptr(), R, gb.data_ptr(), _ld(gb),
                                  gb.data_ptr() + 4 * c, _ld(gb), y.data_ptr(), c, n * h * w, c, act, param, _stream()),
              "spade_norm_fwd")
        ctx.save_for_backward(x, gb, mean, rstd)
        ctx.cfg = (G, R, act, param)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = lib()
        x, gb, mean, rstd = ctx.saved_tensors
        G, R, act, param = ctx.cfg
        dy = to_rows(dy)
        n, c, h, w = x.shape
        rows = n * h * w
        dgb = nhwc_empty(n, h, w, 2 * c, x.device)
        dn = nhwc_empty(n, h, w, c, x.device)
        nb = L.so_spade_bwd_colsum_blocks(rows, c) if _ld(x) % 4 == 0 and _ld(gb) % 4 == 0 and _ld(dy) % 4 == 0 else 0
        part = torch.empty((nb, 2 * c), dtype=torch.float32, device=x.device) if nb else None
        check(L.so_spade_norm_bwd(x.data_ptr(), _ld(x), mean.data_ptr(), rstd.data_ptr(), R, gb.data_ptr(), _ld(gb),
                                  gb.data_ptr() + 4 * c, _ld(gb), dy.data_ptr(), _ld(dy), dn.data_ptr(), c, dgb.data_ptr(), 2 * c,
                                  dgb.data_ptr() + 4 * c, 2 * c, rows, c, act, param, part.data_ptr() if nb else None, _stream()),
              "spade_norm_bwd")
        if nb:
            db = torch.empty(2 * c, dtype=torch.float32, device=x.device)
            ws = workspace(x.device, L.so_colsum_ws_floats(nb, 2 * c) * 4, lane=6)
            check(L.so_colsum(part.data_ptr(), 2 * c, nb, 2 * c, db.data_ptr(), 0, ws.data_ptr(), _stream()), "colsum")
            dgb._so_bias_grad = db
        dx = None
        if ctx.needs_input_grad[0]:
            dx = nhwc_empty(n, h, w, c, x.device)
            ws = workspace(x.device, L.so_norm_ws_floats(G, R, c) * 4)
            check(L.so_norm_bwd(x.data_ptr(), _ld(x), dn.data_ptr(), c, dx.data_ptr(), c, G, R, c, mean.data_ptr(), rstd.data_ptr(),
                                None, None, None, 0, 0, ws.data_ptr(), _stream()), "norm_bwd")
        return dx, dgb, None, None, None, None, None, None, None


def spade_norm_modulate(x, gamma_beta, running_mean=None, running_var=None, instance=False, momentum=0.1, eps=1e-5,
                        act=ACT_NONE, param=0.0):
    """SPADE.forward in training mode (sams/spade.py:80-89): batch statistics (instance=False; the running statistics get
    their momentum update) or per-sample statistics (instance=True), modulation and the following activation."""
    _require_cuda(x)
    if gamma_beta.shape[1] != 2 * x.shape[1]:
        raise ValueError("gamma_beta must hold 2 * C channels")
    return _SpadeNormFn.apply(x, gamma_beta, running_mean, running_var, bool(instance), float(momentum), float(eps),
                              ACT_CODES[act] if not isinstance(act, int) else act, float(param))


class _StackConvParamsFn(torch.autograd.Function):
    """(w_a, b_a, w_b, b_b) -> (cat([w_a, w_b], 0), cat([b_a, b_b])) in OHWI memory.  The gradient of the stacked tensors
    is handed back as two row blocks; parameters whose .grad lives in the optimizer's flat slab are accumulated in place."""

    @staticmethod
    def forward(ctx, wa, ba, wb, bb):
        L = lib()
        o, i, r, s = wa.shape
        w2 = torch.empty((2 * o, r, s, i), dtype=torch.float32, device=wa.device)
        b2 = torch.empty(2 * o, dtype=torch.float32, device=wa.device)
        for k, (w, b) in enumerate(((wa, ba), (wb, bb))):
            wd = _ohwi_dense(w.detach())
            check(L.so_copy2d(wd.data_ptr(), i, i, w2.data_ptr() + 4 * k * o * r * s * i, i, i, o * r * s, 0, _stream()), "copy2d")
            check(L.so_copy2d(b.data_ptr(), o, o, b2.data_ptr() + 4 * k * o, o, o, 1, 0, _stream()), "copy2d")
        ctx.o = o
        ctx.direct = tuple(p if _direct_grad_ok(p, ohwi=(p.dim() == 4)) else None for p in (wa, ba, wb, bb))
        return w2.permute(0, 3, 1, 2), b2

    @staticmethod
    def backward(ctx, gw, gb):
        L = lib()
        o = ctx.o
        gw = _ohwi_dense(gw)
        gb = gb.contiguous()
        _, r, s, i = gw.shape
        outs = []
        for k in range(2):
            wpart = gw[k * o:(k + 1) * o]
            bpart = gb[k * o:(k + 1) * o]
            pw, pb = ctx.direct[2 * k], ctx.direct[2 * k + 1]
            if pw is not None:
                check(L.so_copy2d(wpart.data_ptr(), i, i, pw.grad.data_ptr(), i, i, o * r * s, 1, _stream()), "copy2d")
                outs.append(None)
            else:
                outs.append(wpart.permute(0, 3, 1, 2))
            if pb is not None:
                check(L.so_axpby(bpart.data_ptr(), 1.0, pb.grad.data_ptr(), 1.0, o, _stream()), "axpby")
                outs.append(None)
            else:
                outs.append(bpart)
            grad_ready(pw, pb)
        return tuple(outs)


def stack_conv_params(wa, ba, wb, bb):
    """Two convolutions that read the same input (SPADE's mlp_gamma / mlp_beta, sams/spade.py:85-86) as one with 2 * O
    output channels: returns the stacked (weight, bias)."""
    return _StackConvParamsFn.apply(wa, ba, wb, bb)


# ------------------------------------------------------------------------------------------------
# average pool between discriminator scales
# ------------------------------------------------------------------------------------------------
class _AvgPool3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = to_rows(x)
        n, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = nhwc_empty(n, ho, wo, c, x.device)
        check(lib().so_avgpool3s2_fwd(x.data_ptr(), _ld(x), y.data_ptr(), c, n, h, w, c, _stream()), "avgpool3s2_fwd")
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        dy = to_rows(dy)
        dx = nhwc_empty(n, h, w, c, dy.device)
        check(lib().so_avgpool3s2_bwd(dy.data_ptr(), _ld(dy), dx.data_ptr(), c, n, h, w, c, _stream()), "avgpool3s2_bwd")
        return dx


def avg_pool3s2(x):
    """F.avg_pool2d(x, kernel_size=3, stride=2, padding=1, count_include_pad=False) (discriminator.py:51-54)."""
    _require_cuda(x)
    return _AvgPool3s2Fn.apply(x)


# ------------------------------------------------------------------------------------------------
# spectral norm
# ------------------------------------------------------------------------------------------------
class _SpectralNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight_orig, uv, power_iter, eps):
        L = lib()
        u, v = uv
        o, i, r, s = weight_orig.shape
        w = _ohwi_dense(weight_orig.detach())
        out = torch.empty((o, r, s, i), dtype=torch.float32, device=w.device)
        sigma = torch.empty(1, dtype=torch.float32, device=w.device)
        ws = workspace(w.device, L.so_spectral_norm_ws_floats(o, i, r * s) * 4, lane=6)
        check(L.so_spectral_norm_fwd(w.data_ptr(), o, i, r * s, u.data_ptr(), v.data_ptr(), out.data_ptr(), sigma.data_ptr(),
                                     int(power_iter), eps, ws.data_ptr(), _stream()), "spectral_norm_fwd")
        # u / v are overwritten by the next forward (one power iteration per call, five calls per generator step):
        # the backward pass needs the pair this sigma was computed with
        ctx.save_for_backward(w, u.clone() if power_iter else u, v.clone() if power_iter else v, sigma)
        ctx.direct = weight_orig if _direct_grad_ok(weight_orig, ohwi=True) else None
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        L = lib()
        w, u, v, sigma = ctx.saved_tensors
        o, r, s, i = w.shape
        g = _ohwi_dense(g)
        ws = workspace(w.device, L.so_spectral_norm_ws_floats(o, i, r * s) * 4, lane=6)
        if ctx.direct is not None:
            check(L.so_spectral_norm_bwd(g.data_ptr(), w.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr(), o, i, r * s,
                                         ctx.direct.grad.data_ptr(), 1, ws.data_ptr(), _stream()), "spectral_norm_bwd")
            grad_ready(ctx.direct)
            return None, None, None, None
        dw = torch.empty((o, r, s, i), dtype=torch.float32, device=w.device)
        check(L.so_spectral_norm_bwd(g.data_ptr(), w.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr(), o, i, r * s,
                                     dw.data_ptr(), 0, ws.data_ptr(), _stream()), "spectral_norm_bwd")
        return dw.permute(0, 3, 1, 2), None, None, None


def spectral_normalize(weight_orig, u, v, training, eps=1e-12):
    """torch.nn.utils.spectral_norm's weight computation: in training mode ONE power iteration updates the buffers `u`
    and `v` in place (also under torch.no_grad(), as the reference's hook does), then W / (u^T W v)."""
    _require_cuda(weight_orig)
    return _SpectralNormFn.apply(weight_orig, (u, v), bool(training), float(eps))


# ------------------------------------------------------------------------------------------------
# GAN losses
# ------------------------------------------------------------------------------------------------
class _GanLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode, real, for_disc):
        x = to_rows(x)
        n, c, h, w = x.shape
        out = torch.empty((), dtype=torch.float32, device=x.device)
        ws = workspace(x.device, 4096, lane=6)
        check(lib().so_gan_loss_fwd(x.data_ptr(), _ld(x), n * h * w, c, mode, real, for_disc, out.data_ptr(), ws.data_ptr(),
                                    _stream()), "gan_loss_fwd")
        ctx.save_for_backward(x)
        ctx.cfg = (mode, real, for_disc)
        return out

    @staticmethod
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        mode, real, for_disc = ctx.cfg
        n, c, h, w = x.shape
        gout = gout.contiguous()
        dx = nhwc_empty(n, h, w, c, x.device)
        check(lib().so_gan_loss_bwd(x.data_ptr(), _ld(x), n * h * w, c, mode, real, for_disc, gout.data_ptr(), dx.data_ptr(), c,
                                    _stream()), "gan_loss_bwd")
        return dx, None, None, None


def gan_loss(x, mode, target_is_real, for_discriminator=True):
    """GANLoss.loss (models/networks/loss.py:58-88) of one prediction tensor -> 0-d tensor."""
    _require_cuda(x)
    if mode == "hinge" and not for_discriminator and not target_is_real:
        raise AssertionError("The generator's hinge loss must be aiming for real")
    return _GanLossFn.apply(x, GAN_MODES[mode], int(bool(target_is_real)), int(bool(for_discriminator)))


from .ops import make_functions_lane_aware  # noqa: E402

make_functions_lane_aware(globals())

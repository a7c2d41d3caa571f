"""GPU parity of every HIP operator (through the C ABI) against a plain PyTorch fp32 CPU reference of
the same op / the oracle, forward and backward, on seeded inputs.  Tolerances are written per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close, oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(cuda):
    from shineon_virtual_tryon_amd import ops as _ops

    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(tuple(shape), generator=g) * scale


def grads(outs, inputs, seeds):
    """d(sum_i <out_i, seed_i>) / d inputs."""
    loss = sum((o * s).sum() for o, s in zip(outs, seeds))
    return torch.autograd.grad(loss, inputs, allow_unused=True)


def compare_fwd_bwd(hip_fn, ref_fn, inputs, cuda, atol=1e-4, rtol=1e-4, gatol=None, what="", grel=None):
    """inputs: list of (tensor, requires_grad).  Functions return a tensor or tuple of tensors."""
    gatol = atol if gatol is None else gatol
    cpu_in = [t.clone().requires_grad_(rg) for t, rg in inputs]
    gpu_in = [t.clone().to(cuda).requires_grad_(rg) for t, rg in inputs]
    r = ref_fn(*cpu_in)
    h = hip_fn(*gpu_in)
    r = r if isinstance(r, (tuple, list)) else (r,)
    h = h if isinstance(h, (tuple, list)) else (h,)
    for i, (a, b) in enumerate(zip(h, r)):
        assert_close(a, b, atol=atol, rtol=rtol, what=f"{what} out[{i}]")
    seeds = [rnd(*o.shape, seed=100 + i) for i, o in enumerate(r)]
    need = [x for x, (_, rg) in zip(cpu_in, inputs) if rg]
    if not need:
        return
    gr = grads(r, need, seeds)
    gh = grads(h, [x for x, (_, rg) in zip(gpu_in, inputs) if rg], [s.to(cuda) for s in seeds])
    for i, (a, b) in enumerate(zip(gh, gr)):
        # grel: tolerance relative to the largest reference gradient entry (long reductions, e.g. TPS over H*W)
        tol = gatol if grel is None else grel * b.abs().max().item()
        assert_close(a, b, atol=tol, rtol=rtol, what=f"{what} grad[{i}]")


# ------------------------------------------------------------------------------------------------ conv
CONV_CASES = [
    # (N, Cin, H, W, Cout, k, stride, pad)
    (2, 12, 20, 12, 64, 4, 2, 1),
    (2, 10, 16, 12, 40, 4, 2, 1),      # Cin not a multiple of 4 -> zero-padded channel path
    (3, 64, 12, 10, 36, 3, 1, 1),
    (1, 128, 8, 6, 128, 3, 1, 1),
    (2, 32, 9, 7, 16, 1, 1, 0),
    (2, 3, 16, 12, 64, 3, 1, 1),       # VGG conv1_1 shape class
    (4, 512, 4, 3, 512, 4, 2, 1),      # innermost U-Net level: tiny M, K = 8192
    (2, 64, 16, 12, 4, 3, 1, 1),       # four output channels: 4x4x1-MFMA kernels (fprop + wgrad), U-Net last conv
    (2, 128, 9, 7, 3, 3, 1, 1),        # three output channels padded to four, ragged pixel count
    (2, 4, 16, 12, 64, 3, 1, 1),       # four input channels: 4x4x1-MFMA input gradient (VGG conv1_1 class)
    (3, 64, 8, 6, 4, 1, 1, 0),         # 1x1 with four output channels
    (2, 40, 9, 7, 4, 3, 1, 1),         # four output channels, 10 channel quads: ragged tail of the 4-quad prefetch groups
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("force", [(0, 0, 0), (128, 128, 1), (64, 64, 1), (64, 64, 4), (128, 128, 2), (128, 64, 1), (64, 128, 2)])
def test_conv2d_fwd_bwd(ops, cuda, case, force):
    from shineon_virtual_tryon_amd import lib

    n, ci, h, w, co, k, s, p = case
    x = rnd(n, ci, h, w, seed=1)
    wt = rnd(co, ci, k, k, seed=2, scale=(2.0 / (ci * k * k)) ** 0.5)
    b = rnd(co, seed=3, scale=0.1)
    lib().so_igemm_force(*force)
    try:
        compare_fwd_bwd(
            lambda x_, w_, b_: ops.conv2d(x_, w_, b_, s, p),
            lambda x_, w_, b_: F.conv2d(x_, w_, b_, stride=s, padding=p),
            [(x, ci % 4 == 0), (wt, True), (b, True)], cuda, atol=2e-5, rtol=1e-4, gatol=1e-4,
            what=f"conv {case} force={force}",
        )
    finally:
        lib().so_igemm_force(0, 0, 0)


@pytest.mark.parametrize("waves", [8])
def test_conv2d_other_wave_counts(ops, cuda, waves):
    """The 8-wave 128x128 tile, forced (the measured plans pick it per shape)."""
    from shineon_virtual_tryon_amd import lib

    n, ci, h, w, co, k, s, p = 2, 32, 12, 10, 96, 3, 1, 1
    x, wt, b = rnd(n, ci, h, w, seed=50), rnd(co, ci, k, k, seed=51, scale=0.06), rnd(co, seed=52, scale=0.1)
    lib().so_igemm_force(64 if waves == 2 else 128, 64 if waves == 2 else 128, 2)
    lib().so_igemm_force_waves(waves)
    try:
        compare_fwd_bwd(lambda x_, w_, b_: ops.conv2d(x_, w_, b_, s, p), lambda x_, w_, b_: F.conv2d(x_, w_, b_, stride=s, padding=p),
                        [(x, True), (wt, True), (b, True)], cuda, atol=2e-5, rtol=1e-4, gatol=1e-4, what=f"conv waves={waves}")
    finally:
        lib().so_igemm_force(0, 0, 0)


def test_conv2d_fused_relu(ops, cuda):
    x, wt, b = rnd(2, 16, 10, 8, seed=4), rnd(24, 16, 3, 3, seed=5, scale=0.1), rnd(24, seed=6, scale=0.1)
    compare_fwd_bwd(lambda x_, w_, b_: ops.conv2d(x_, w_, b_, 1, 1, ops.ACT_RELU),
                    lambda x_, w_, b_: F.relu(F.conv2d(x_, w_, b_, padding=1)),
                    [(x, True), (wt, True), (b, True)], cuda, atol=2e-5, what="conv+relu")


@pytest.mark.parametrize("kind,ref", [("gelu", F.gelu), ("tanh", torch.tanh), ("swish", lambda t: t * torch.sigmoid(t))])
@pytest.mark.parametrize("case", [(2, 16, 10, 8, 24, 3, 1, 1), (2, 64, 16, 12, 4, 3, 1, 1), (4, 512, 4, 3, 512, 4, 2, 1)])
def test_conv2d_forward_with_other_activations(ops, cuda, kind, ref, case):
    """C-ABI contract of so_conv2d_fprop's `act`: the engine's epilogues fuse none / ReLU / LeakyReLU; every other SO_ACT_* is
    applied as a second pass over the output (incl. the split-K and the four-output-channel paths) - same values as fused."""
    from shineon_virtual_tryon_amd import lib

    n, ci, h, w, co, k, st, p = case
    x, wt, b = rnd(n, ci, h, w, seed=31), rnd(co, ci, k, k, seed=32, scale=(2.0 / (ci * k * k)) ** 0.5), rnd(co, seed=33, scale=0.1)
    xg = x.permute(0, 2, 3, 1).contiguous().to(cuda)          # NHWC
    wg = wt.permute(0, 2, 3, 1).contiguous().to(cuda)         # OHWI
    bg = b.to(cuda)
    ho, wo = (h + 2 * p - k) // st + 1, (w + 2 * p - k) // st + 1
    y = torch.empty(n, ho, wo, co, device=cuda)
    ws = ops.workspace(cuda)
    err = lib().so_conv2d_fprop(xg.data_ptr(), ci, wg.data_ptr(), bg.data_ptr(), y.data_ptr(), co, n, h, w, ci, co, k, k, st, p,
                                ops.ACT_CODES[kind], 0.0, ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
    assert err == 0
    assert_close(y.permute(0, 3, 1, 2), ref(F.conv2d(x, wt, b, stride=st, padding=p)), atol=2e-5, what=f"conv+{kind} {case}")


def test_conv2d_channel_slice_input(ops, cuda):
    """Operand that is a channel slice (pitch > C) of a wider NHWC buffer."""
    full = rnd(2, 24, 8, 6, seed=7)
    wt = rnd(8, 16, 3, 3, seed=8, scale=0.1)
    fg = ops.to_rows(full.to(cuda))
    y = ops.conv2d(fg[:, 8:24], wt.to(cuda), None, 1, 1)
    assert_close(y, F.conv2d(full[:, 8:24], wt, padding=1), atol=2e-5, what="sliced conv")


@pytest.mark.parametrize("K", [36, 40, 192])   # reduction lengths of one, two and six 32-wide k-steps (36, 40: ragged last step)
@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0)])
def test_gemm_batched(ops, cuda, ta, tb, K):
    from shineon_virtual_tryon_amd import lib

    B, M, N = 3, 52, 44
    a = rnd(B, K, M, seed=9) if ta else rnd(B, M, K, seed=9)
    b = rnd(B, N, K, seed=10) if tb else rnd(B, K, N, seed=10)
    ref = torch.bmm(a.transpose(1, 2) if ta else a, b.transpose(1, 2) if tb else b)
    ag, bg = a.to(cuda), b.to(cuda)
    c = torch.empty(B, M, N, device=cuda)
    ws = ops.workspace(cuda)
    lda, ldb = a.shape[2], b.shape[2]
    err = lib().so_gemm_batched(ta, tb, M, N, K, ag.data_ptr(), lda, a[0].numel(), bg.data_ptr(), ldb, b[0].numel(),
                                c.data_ptr(), N, M * N, B, None, None, None, 0, 0, 0, 0.0, ws.data_ptr(), ws.numel() * 4,
                                torch.cuda.current_stream().cuda_stream)
    assert err == 0
    # entries are sums of K unit-variance products (|c| ~ sqrt(K)): the fp32 summation-order difference grows with K
    assert_close(c, ref, atol=2e-5 if K <= 40 else 1e-4, what=f"gemm ta={ta} tb={tb} K={K}")


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0)])
@pytest.mark.parametrize("relu", [0, 1])
def test_gemm_batched_epilogue(ops, cuda, ta, tb, relu):
    """alpha, bias, residual and ReLU in the batched GEMM's epilogue; ragged edge tiles in M and N."""
    from shineon_virtual_tryon_amd import lib

    B, M, N, K = 2, 76, 100, 64
    a = rnd(B, K, M, seed=11) if ta else rnd(B, M, K, seed=11)
    b = rnd(B, N, K, seed=12) if tb else rnd(B, K, N, seed=12)
    alpha, bias, res = torch.tensor([0.37]), rnd(N, seed=13), rnd(B, M, N, seed=14)
    ref = 0.37 * torch.bmm(a.transpose(1, 2) if ta else a, b.transpose(1, 2) if tb else b) + bias + res
    ref = F.relu(ref) if relu else ref
    ag, bg, alg, big, rg = (t.to(cuda) for t in (a, b, alpha, bias, res))
    c = torch.empty(B, M, N, device=cuda)
    ws = ops.workspace(cuda)
    err = lib().so_gemm_batched(ta, tb, M, N, K, ag.data_ptr(), a.shape[2], a[0].numel(), bg.data_ptr(), b.shape[2], b[0].numel(),
                                c.data_ptr(), N, M * N, B, alg.data_ptr(), big.data_ptr(), rg.data_ptr(), N, M * N, relu, 0.0,
                                ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
    assert err == 0
    assert_close(c, ref, atol=2e-5, what=f"gemm epilogue ta={ta} tb={tb} relu={relu}")


# ------------------------------------------------------------------------------------------------ pointwise
@pytest.mark.parametrize("kind,ref", [
    ("relu", F.relu), ("leaky", lambda t: F.leaky_relu(t, 0.2)), ("gelu", lambda t: F.gelu(t)),
    ("swish", lambda t: t * torch.sigmoid(t)), ("sine", lambda t: torch.sin(30 * t)),
    ("tanh", torch.tanh), ("sigmoid", torch.sigmoid)])
@pytest.mark.parametrize("c", [8, 5])
def test_activation(ops, cuda, kind, ref, c):
    x = rnd(2, c, 7, 5, seed=11)
    tol = 2e-4 if kind == "sine" else 1e-5  # sin(30x): argument error is amplified 30x
    compare_fwd_bwd(lambda t: ops.activation(t, kind, 0.2 if kind == "leaky" else 0.0), ref, [(x, True)], cuda,
                    atol=tol, gatol=30 * tol if kind == "sine" else 1e-5, what=kind)


def test_instance_norm(ops, cuda):
    for shape in ((2, 8, 16, 12), (2, 4, 64, 48), (3, 512, 4, 3), (2, 5, 6, 4)):
        x = rnd(*shape, seed=12) * 2 + 3.0
        compare_fwd_bwd(ops.instance_norm, lambda t: F.instance_norm(t, eps=1e-5), [(x, True)], cuda, atol=2e-5,
                        gatol=5e-5, what=f"instance_norm {shape}")


@pytest.mark.parametrize("shape", [(2, 64, 16, 12), (2, 128, 64, 48), (1, 8, 5, 7)])
@pytest.mark.parametrize("kind", ["gelu", "leaky"])
def test_instance_norm_with_the_consumers_activation_is_bit_identical(ops, cuda, shape, kind):
    """ops.instance_norm_act: (y, act(y)) from ONE launch - values and gradients bit-identical to instance_norm followed by
    activation (the activation's backward node is the same kernel; only its forward launch is gone)."""
    x = rnd(*shape, seed=71).to(cuda)
    gy, ga = rnd(*shape, seed=72).to(cuda), rnd(*shape, seed=73).to(cuda)
    x1 = x.clone().requires_grad_(True)
    y1 = ops.instance_norm(x1, 1e-5)
    a1 = ops.activation(y1, kind, 0.2)
    (g1,) = torch.autograd.grad((y1 * gy).sum() + (a1 * ga).sum(), x1)
    x2 = x.clone().requires_grad_(True)
    y2, a2 = ops.instance_norm_act(x2, 1e-5, kind, 0.2)
    (g2,) = torch.autograd.grad((y2 * gy).sum() + (a2 * ga).sum(), x2)
    assert torch.equal(y1, y2) and torch.equal(a1, a2) and torch.equal(g1, g2)


@pytest.mark.parametrize("shape", [(2, 64, 16, 12), (1, 6, 5, 7)])   # 16-byte path / scalar path
@pytest.mark.parametrize("kind", ["gelu", "leaky", "relu"])
def test_fork_act_is_bit_identical_to_activation_plus_autograd_accumulation(ops, cuda, shape, kind):
    """ops.fork_act: (x, act(x)) as ONE node (a U-Net block input feeds the skip concatenation and the down activation,
    unet.py:187-198); its backward kernel dx = d_skip + d_act * act'(x) must give the bits of act_bwd followed by autograd's
    accumulation add, and cope with either gradient missing."""
    x = rnd(*shape, seed=81).to(cuda)
    gs, ga = rnd(*shape, seed=82).to(cuda), rnd(*shape, seed=83).to(cuda)
    x1 = x.clone().requires_grad_(True)
    a1 = ops.activation(x1, kind, 0.2)
    (g1,) = torch.autograd.grad((x1 * gs).sum() + (a1 * ga).sum(), x1)
    x2 = x.clone().requires_grad_(True)
    xs, a2 = ops.fork_act(x2, kind, 0.2)
    (g2,) = torch.autograd.grad((xs * gs).sum() + (a2 * ga).sum(), x2)
    assert torch.equal(xs, x) and torch.equal(a1, a2) and torch.equal(g1, g2)
    # only the activation branch / only the skip branch carries a gradient
    x3 = x.clone().requires_grad_(True)
    _, a3 = ops.fork_act(x3, kind, 0.2)
    (g3,) = torch.autograd.grad((a3 * ga).sum(), x3)
    (g3_ref,) = torch.autograd.grad((ops.activation(x1, kind, 0.2) * ga).sum(), x1)
    assert torch.equal(g3, g3_ref)
    x4 = x.clone().requires_grad_(True)
    xs4, _ = ops.fork_act(x4, kind, 0.2)
    (g4,) = torch.autograd.grad((xs4 * gs).sum(), x4)
    assert torch.equal(g4, gs)


def test_instance_norm_known_answer(ops, cuda):
    y = ops.instance_norm(rnd(2, 8, 32, 24, seed=13).to(cuda) * 5 + 1)
    yc = ops.to_nchw(y).cpu()
    assert yc.mean(dim=(2, 3)).abs().max() < 1e-5
    assert (yc.var(dim=(2, 3), unbiased=False) - 1).abs().max() < 1e-3


@pytest.mark.parametrize("hw", [(8, 6), (24, 16)])  # single-launch path (R <= 1024 rows) and the three-kernel path
def test_batch_norm_train_and_eval(ops, cuda, hw):
    x = rnd(4, 16, hw[0], hw[1], seed=14) * 1.5 + 0.5
    g, b = rnd(16, seed=15) * 0.1 + 1, rnd(16, seed=16) * 0.1
    rm, rv = torch.zeros(16), torch.ones(16)
    rm_g, rv_g = rm.clone().to(cuda), rv.clone().to(cuda)
    rm_c, rv_c = rm.clone(), rv.clone()
    compare_fwd_bwd(lambda x_, g_, b_: ops.batch_norm_train(x_, g_, b_, rm_g, rv_g, 0.1, 1e-5),
                    lambda x_, g_, b_: F.batch_norm(x_, rm_c, rv_c, g_, b_, True, 0.1, 1e-5),
                    [(x, True), (g, True), (b, True)], cuda, atol=2e-5, gatol=5e-5, what="batch_norm")
    assert_close(rm_g, rm_c, atol=1e-6, what="running_mean")
    assert_close(rv_g, rv_c, atol=1e-6, what="running_var")
    y = ops.batch_norm_eval(x.to(cuda), g.to(cuda), b.to(cuda), rm_g, rv_g)
    assert_close(y, F.batch_norm(x, rm_c, rv_c, g, b, False, 0.1, 1e-5), atol=2e-5, what="batch_norm eval")


@pytest.mark.parametrize("hw", [(8, 6), (40, 28)])  # single-launch norm (R <= 1024 rows) / three-kernel norm
def test_conv_relu_batchnorm_fused_gate(ops, cuda, hw):
    """FeatureExtraction's Conv -> ReLU -> BatchNorm group: ReLU in the conv epilogue, its backward mask applied by the
    BatchNorm backward (conv2d(act_grad_external=True) + batch_norm_train(relu_gate_input=True))."""
    x = rnd(2, 8, hw[0], hw[1], seed=40)
    w, b = rnd(16, 8, 3, 3, seed=41, scale=0.15), rnd(16, seed=42, scale=0.1)
    g, be = rnd(16, seed=43) * 0.1 + 1, rnd(16, seed=44) * 0.1
    rm_g, rv_g = torch.zeros(16).to(cuda), torch.ones(16).to(cuda)
    rm_c, rv_c = torch.zeros(16), torch.ones(16)

    def hip(x_, w_, b_, g_, be_):
        y = ops.conv2d(x_, w_, b_, 1, 1, ops.ACT_RELU, act_grad_external=True)
        return ops.batch_norm_train(y, g_, be_, rm_g, rv_g, 0.1, 1e-5, relu_gate_input=True)

    compare_fwd_bwd(hip, lambda x_, w_, b_, g_, be_: F.batch_norm(F.relu(F.conv2d(x_, w_, b_, padding=1)), rm_c, rv_c, g_, be_,
                                                                   True, 0.1, 1e-5),
                    [(x, True), (w, True), (b, True), (g, True), (be, True)], cuda, atol=2e-5, gatol=1e-4, what=f"conv+relu+bn {hw}")


@pytest.mark.parametrize("hw", [(8, 6), (40, 28)])  # single-launch norm (R <= 1024 rows) / three-kernel norm
def test_conv_bias_gradient_from_the_batchnorm_backward(ops, cuda, hw):
    """Conv -> ReLU -> BatchNorm with the parameters in the optimizer's slab (FeatureExtraction.forward, warp.py:15-31): the
    convolution's bias gradient - the column sums of the gated gradient the BatchNorm returns - comes out of the BatchNorm's
    backward statistics pass (so_norm_bwd_bias: summed in the one-launch kernel, analytically from five column sums in the
    three-kernel path) and the convolution computes none.  Against torch, and against the path that sums the columns of dx."""
    from shineon_virtual_tryon_amd.optim import HipAdam

    x = rnd(2, 8, hw[0], hw[1], seed=50)
    seed_out = rnd(2, 16, hw[0], hw[1], seed=55)
    vals = dict(w=rnd(16, 8, 3, 3, seed=51, scale=0.15), b=rnd(16, seed=52, scale=0.1), g=rnd(16, seed=53) * 0.1 + 1,
                be=rnd(16, seed=54) * 0.1)
    got = {}
    for ext in (True, False):
        prm = {k: torch.nn.Parameter(v.clone().to(cuda)) for k, v in vals.items()}
        opt = HipAdam(list(prm.values()), lr=1e-3)
        opt.zero_grad()
        assert ops._direct_grad_ok(prm["b"], False)
        rm, rv = torch.zeros(16, device=cuda), torch.ones(16, device=cuda)
        y = ops.conv2d(x.to(cuda), prm["w"], prm["b"], 1, 1, ops.ACT_RELU, act_grad_external=True, bias_grad_external=ext)
        out = ops.batch_norm_train(y, prm["g"], prm["be"], rm, rv, 0.1, 1e-5, relu_gate_input=True,
                                   conv_bias=prm["b"] if ext else None)
        (out * seed_out.to(cuda)).sum().backward()
        got[ext] = {k: v.grad.detach().clone().cpu() for k, v in prm.items()}
    ref = {k: v.clone().requires_grad_(True) for k, v in vals.items()}
    o = F.batch_norm(F.relu(F.conv2d(x, ref["w"], ref["b"], padding=1)), torch.zeros(16), torch.ones(16), ref["g"], ref["be"],
                     True, 0.1, 1e-5)
    (o * seed_out).sum().backward()
    for k in vals:
        assert_close(got[True][k], ref[k].grad, atol=1e-4,
                     what=f"d{k} with the bias gradient from the BatchNorm backward {hw}")
        assert_close(got[True][k], got[False][k], atol=1e-4 if k == "b" else 0.0, what=f"d{k}: BatchNorm-side vs column sums {hw}")


@pytest.mark.parametrize("kind,ref", [("relu", F.relu), ("gelu", F.gelu), ("leaky", lambda t: F.leaky_relu(t, 0.2))])
@pytest.mark.parametrize("c", [8, 6])
def test_activation_fused_into_upsample(ops, cuda, kind, ref, c):
    """The U-Net up path's `act -> Upsample(x2, bilinear)` pair as one kernel (forward and backward)."""
    x = rnd(2, c, 6, 5, seed=18)
    compare_fwd_bwd(lambda t: ops.upsample2x_bilinear(t, kind, 0.2 if kind == "leaky" else 0.0),
                    lambda t: F.interpolate(ref(t), scale_factor=2, mode="bilinear", align_corners=False),
                    [(x, True)], cuda, atol=2e-6, what=f"{kind}+upsample")


@pytest.mark.parametrize("c1,c2", [(8, 12), (6, 5)])  # 16-byte path / scalar path
@pytest.mark.parametrize("kind,ref", [("gelu", F.gelu), (None, lambda t: t)])
def test_upsample_of_unmaterialised_concat(ops, cuda, c1, c2, kind, ref):
    """upsample(act(cat([a, b], 1))) reading both parts in place; gradients written to the two inputs separately."""
    a, b = rnd(2, c1, 6, 5, seed=19), rnd(2, c2, 6, 5, seed=20)
    compare_fwd_bwd(lambda a_, b_: ops.upsample2x_bilinear_cat(a_, b_, kind),
                    lambda a_, b_: F.interpolate(ref(torch.cat([a_, b_], 1)), scale_factor=2, mode="bilinear", align_corners=False),
                    [(a, True), (b, True)], cuda, atol=2e-6, what=f"cat+{kind}+upsample")


@pytest.mark.parametrize("kind,ref", [("gelu", F.gelu), (None, lambda t: t)])
@pytest.mark.parametrize("hw", [(28, 21), (32, 24), (19, 40)])   # edge tiles in both directions / exact tiles
def test_upsample_tiled_levels(ops, cuda, kind, ref, hw):
    """The LDS-tiled kernels of the large levels (>= 512 source pixels, channels in chunks of 32): single source and the
    unmaterialised concatenation, forward and backward, against torch."""
    h, w = hw
    x = rnd(2, 64, h, w, seed=23)
    compare_fwd_bwd(lambda t: ops.upsample2x_bilinear(t, kind),
                    lambda t: F.interpolate(ref(t), scale_factor=2, mode="bilinear", align_corners=False),
                    [(x, True)], cuda, atol=2e-6, what=f"tiled {kind}+upsample {hw}")
    a, b = rnd(2, 32, h, w, seed=24), rnd(2, 64, h, w, seed=25)
    compare_fwd_bwd(lambda a_, b_: ops.upsample2x_bilinear_cat(a_, b_, kind),
                    lambda a_, b_: F.interpolate(ref(torch.cat([a_, b_], 1)), scale_factor=2, mode="bilinear", align_corners=False),
                    [(a, True), (b, True)], cuda, atol=2e-6, what=f"tiled cat+{kind}+upsample {hw}")


def test_upsample_maxpool_cat(ops, cuda):
    x = rnd(2, 8, 6, 5, seed=17)
    compare_fwd_bwd(ops.upsample2x_bilinear, lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False),
                    [(x, True)], cuda, atol=1e-6, what="upsample")
    x1 = rnd(2, 3, 1, 1, seed=18)
    compare_fwd_bwd(ops.upsample2x_bilinear, lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False),
                    [(x1, True)], cuda, atol=1e-6, what="upsample 1x1")
    xp = rnd(2, 8, 8, 6, seed=19)
    compare_fwd_bwd(ops.maxpool2x2, lambda t: F.max_pool2d(t, 2, 2), [(xp, True)], cuda, atol=0, rtol=0, what="maxpool")
    a, b = rnd(2, 8, 4, 3, seed=20), rnd(2, 5, 4, 3, seed=21)
    compare_fwd_bwd(lambda p, q: ops.cat_channels([p, q]), lambda p, q: torch.cat([p, q], 1), [(a, True), (b, True)],
                    cuda, atol=0, rtol=0, what="cat")


def test_layout_roundtrip(ops, cuda):
    x = rnd(2, 7, 5, 3, seed=22)
    r = ops.to_rows(x.to(cuda))
    assert r.shape == x.shape and r.stride()[1] == 1
    assert torch.equal(ops.to_nchw(r).cpu(), x)
    rp = ops.to_rows(x.to(cuda), cpad=8)
    assert torch.equal(rp[:, :7].cpu(), x) and rp[:, 7:].abs().sum() == 0


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("hw", [(4, 3), (8, 6), (16, 12)])
def test_self_attention(ops, cuda, hw):
    h, w = hw
    c = 64
    x = rnd(2, c, h, w, seed=23)
    wq, wk = rnd(c // 8, c, 1, 1, seed=24, scale=0.05), rnd(c // 8, c, 1, 1, seed=25, scale=0.05)
    wv = rnd(c, c, 1, 1, seed=26, scale=0.1)
    bq, bk, bv = rnd(c // 8, seed=27, scale=0.1), rnd(c // 8, seed=28, scale=0.1), rnd(c, seed=29, scale=0.1)
    gamma = torch.tensor([0.7])

    def ref(x_, wq_, bq_, wk_, bk_, wv_, bv_, g_):
        sd = {"a.query_conv.weight": wq_, "a.query_conv.bias": bq_, "a.key_conv.weight": wk_, "a.key_conv.bias": bk_,
              "a.value_conv.weight": wv_, "a.value_conv.bias": bv_, "a.gamma": g_}
        return oracle.self_attention(x_, sd, "a")

    compare_fwd_bwd(ops.self_attention, ref, [(t, True) for t in (x, wq, bq, wk, bk, wv, bv, gamma)], cuda,
                    atol=2e-5, gatol=1e-4, what=f"self_attention {hw}")
    # forward only (no gradient asked for): the three projections run as one GEMM on a cached [Wq; Wk; Wv] - same result, and
    # the cache follows the parameters (an in-place update bumps the version; a new tensor is a new entry)
    dev_in = [t.to(cuda) for t in (x, wq, bq, wk, bk, wv, bv, gamma)]
    with torch.no_grad():
        want = ref(x, wq, bq, wk, bk, wv, bv, gamma)
        for _ in range(2):   # second call: cache hit
            assert_close(ops.self_attention(*dev_in), want, atol=2e-5, what=f"self_attention forward-only {hw}")
        assert ops._QKV_CACHE, "the forward-only path did not run"
        dev_in[5].mul_(1.5)
        assert_close(ops.self_attention(*dev_in), ref(x, wq, bq, wk, bk, wv * 1.5, bv, gamma), atol=3e-5,
                     what=f"self_attention forward-only after an in-place weight update {hw}")


def test_forward_only_weight_caches_follow_the_parameters(ops, cuda):
    """The gradient-free conv2d path caches the Winograd-domain weights of a stand-alone parameter per (object, version,
    address): an optimizer step (version bump) must be seen, a write through `.data` (no bump) needs
    ops.invalidate_weight_caches(); while a stream captures no entry is created, entries made by eager launches are served and
    pinned for the graph (the frozen VGG's Winograd-domain filters inside the replayed training step)."""
    x = rnd(4, 128, 32, 24, seed=90).to(cuda)      # 3072 pixels x 128 channels: the non-fused Winograd form (ops._wino_mode)
    w = torch.nn.Parameter(rnd(128, 128, 3, 3, seed=91, scale=0.05).to(cuda))
    b = torch.nn.Parameter(rnd(128, seed=92, scale=0.1).to(cuda))
    assert ops._wino_mode(128, 128, 4, 32, 24) != "direct"

    def fresh():
        ops.invalidate_weight_caches()
        with torch.no_grad():
            return ops.to_nchw(ops.conv2d(x, w, b, 1, 1)).clone()

    with torch.no_grad():
        y0 = ops.to_nchw(ops.conv2d(x, w, b, 1, 1)).clone()
        assert ops._WINO_W_CACHE, "the forward-only Winograd path did not run / cache"
        assert torch.equal(ops.to_nchw(ops.conv2d(x, w, b, 1, 1)), y0)            # cache hit
    opt = torch.optim.SGD([w, b], lr=0.5)
    w.grad, b.grad = torch.ones_like(w), torch.ones_like(b)
    opt.step()                                                                       # in-place update: version bump
    with torch.no_grad():
        y1 = ops.to_nchw(ops.conv2d(x, w, b, 1, 1)).clone()
    assert not torch.equal(y1, y0) and torch.equal(y1, fresh())
    w.data.mul_(0.5)                                                                 # through .data: no version bump
    with torch.no_grad():
        stale = ops.to_nchw(ops.conv2d(x, w, b, 1, 1)).clone()
    y2 = fresh()
    assert not torch.equal(y2, y1)
    assert torch.equal(stale, y1), "expected the documented stale hit without invalidate_weight_caches()"
    # under capture no entry is CREATED (its transform would be recorded, not run) ...
    ops.invalidate_weight_caches()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        yc = ops.conv2d(x, w, b, 1, 1)
    assert not ops._WINO_W_CACHE, "a transform recorded into a graph was published to the cache"
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(ops.to_nchw(yc), y2)
    # ... but an entry made by an eager launch is served to a capture, and pinned: the graph must survive the cache being dropped
    with torch.no_grad():
        ops.conv2d(x, w, b, 1, 1)
    torch.cuda.synchronize()
    (entry,) = list(ops._WINO_W_CACHE.values())
    pins = len(ops._CAPTURE_PINS)
    g2 = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g2):
        yc2 = ops.conv2d(x, w, b, 1, 1)
    assert entry[2].data_ptr() in ops._CAPTURE_PINS and len(ops._CAPTURE_PINS) == pins + 1
    gen = ops.cache_generation()
    ops.invalidate_weight_caches()
    assert ops.cache_generation() == gen + 1
    torch.empty(entry[2].numel(), device=cuda).fill_(float("nan"))     # would land in the freed copy if it had been freed
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(ops.to_nchw(yc2), y2)


def test_winograd_fused_block_orders_are_bit_identical(ops, cuda):
    """so_wino_fused_kb_major: the two block orders of the fused kernel (ko block fastest / patch fastest, chosen by filter
    size) only change which L2 an item's operands are fetched into - the results must be the same bits."""
    L = ops.lib()
    st = torch.cuda.current_stream().cuda_stream
    nb, h, w, c, ko = 2, 24, 20, 64, 160      # ragged patches, three 64-channel ko blocks (the last one partial)
    x = rnd(nb, h, w, c, seed=98).to(cuda)
    wt = (rnd(ko, 3, 3, c, seed=99) * 0.05).to(cuda)
    u = torch.empty(L.so_wino_fused_weight_floats(ko, c, 0), device=cuda)
    assert L.so_wino_fused_weights(wt.data_ptr(), u.data_ptr(), ko, ko, c, 0, st) == 0
    outs = []
    try:
        for mode in (0, 1):
            L.so_wino_fused_kb_major(mode)
            for kb32 in (1, 0):
                L.so_wino_fused_force_kb32(kb32)
                y = torch.full((nb, h, w, ko), float("nan"), device=cuda)
                assert L.so_wino_fused_conv3x3(x.data_ptr(), c, u.data_ptr(), None, 0, None, y.data_ptr(), ko, nb, h, w, c, ko, 1, 0.0, st) == 0
                outs.append(y)
    finally:
        L.so_wino_fused_kb_major(-1)
        L.so_wino_fused_force_kb32(-1)
    torch.cuda.synchronize()
    assert not torch.isnan(outs[0]).any()
    for y in outs[1:]:
        assert torch.equal(y, outs[0])


@pytest.mark.parametrize("n_keep", [0, 2, 4])
def test_winograd_fused_conv_with_pooling_epilogue(ops, cuda, n_keep):
    """so_wino_fused_conv3x3_pool (VGG conv1_2 / conv2_2, vgg.py:14-25): conv + bias + ReLU with MaxPool2d(2, 2) written by the
    epilogue, y stored for the first n_keep images only (y may be NULL for n_keep = 0), a pooled pitch wider than Ko, ragged
    Ko; odd H or W is refused.  Against an fp64 convolution."""
    L = ops.lib()
    st = torch.cuda.current_stream().cuda_stream
    nb, h, w, c, ko, ldyp = 4, 16, 24, 32, 40, 48
    x = rnd(nb, c, h, w, seed=95)
    wt = rnd(ko, c, 3, 3, seed=96, scale=(2.0 / (9 * c)) ** 0.5)
    bias = rnd(ko, seed=97, scale=0.1)
    ref = F.relu(F.conv2d(x.double(), wt.double(), bias.double(), padding=1))
    ref_pool = F.max_pool2d(ref, 2, 2)
    xr = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    w_ohwi = wt.permute(0, 2, 3, 1).contiguous().to(cuda)
    u = torch.empty(L.so_wino_fused_weight_floats(ko, c, 0), device=cuda)
    assert L.so_wino_fused_weights(w_ohwi.data_ptr(), u.data_ptr(), ko, ko, c, 0, st) == 0
    y = torch.full((nb, h, w, ko), float("nan"), device=cuda)
    yp = torch.full((nb, h // 2, w // 2, ldyp), float("nan"), device=cuda)
    bdev = bias.to(cuda)
    rc = L.so_wino_fused_conv3x3_pool(xr.data_ptr(), c, u.data_ptr(), bdev.data_ptr(), ko, y.data_ptr() if n_keep else None, ko,
                                      n_keep, yp.data_ptr(), ldyp, nb, h, w, c, ko, 1, 0.0, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    got_pool = yp[..., :ko].permute(0, 3, 1, 2).cpu().double()
    assert (got_pool - ref_pool).abs().max().item() < 2e-5
    assert torch.isnan(yp[..., ko:]).all(), "the pad columns of the pooled map were written"
    if n_keep:
        got = y[:n_keep].permute(0, 3, 1, 2).cpu().double()
        assert (got - ref[:n_keep]).abs().max().item() < 2e-5
        # the pooled map is exactly the maximum of the stored activations
        assert torch.equal(F.max_pool2d(y[:n_keep].permute(0, 3, 1, 2), 2, 2), yp[:n_keep, ..., :ko].permute(0, 3, 1, 2))
    assert torch.isnan(y[n_keep:]).all(), "activations beyond n_keep were stored"
    for hh, ww in ((15, 24), (16, 23)):
        assert L.so_wino_fused_conv3x3_pool(xr.data_ptr(), c, u.data_ptr(), bdev.data_ptr(), ko, y.data_ptr(), ko, nb, yp.data_ptr(),
                                            ldyp, nb, hh, ww, c, ko, 1, 0.0, st) == -2     # SO_ERR_SHAPE


# ------------------------------------------------------------------------------------------------ GMM
def test_l2norm_correlation(ops, cuda):
    fa, fb = rnd(2, 32, 16, 12, seed=30), rnd(2, 32, 16, 12, seed=31)

    def hip(a, b):
        return ops.feature_correlation(ops.feature_l2norm(a, True), ops.feature_l2norm(b, False))

    def ref(a, b):
        return oracle.feature_correlation(oracle.feature_l2norm(a), oracle.feature_l2norm(b))

    compare_fwd_bwd(hip, ref, [(fa, True), (fb, True)], cuda, atol=1e-5, gatol=2e-5, what="l2norm+correlation")


def test_linear_chw_tanh(ops, cuda):
    x, w, b = rnd(4, 64, 4, 3, seed=32), rnd(50, 768, seed=33, scale=0.05), rnd(50, seed=34, scale=0.1)
    compare_fwd_bwd(lambda x_, w_, b_: ops.linear_chw_tanh(x_, w_, b_, True),
                    lambda x_, w_, b_: torch.tanh(F.linear(x_.reshape(4, -1), w_, b_)),
                    [(x, True), (w, True), (b, True)], cuda, atol=1e-5, gatol=2e-5, what="linear+tanh")


def _tps_consts(cuda, h, w, gs):
    c = oracle.tps_constants(h, w, gs)
    return c, tuple(c[k].contiguous().to(cuda) for k in ("Li", "px", "py", "gx", "gy"))


@pytest.mark.parametrize("hw", [(256, 192), (40, 24)])
def test_tps_grid(ops, cuda, hw):
    h, w = hw
    c, dev = _tps_consts(cuda, h, w, 5)
    theta = rnd(2, 50, seed=35, scale=0.15)
    compare_fwd_bwd(lambda t: ops.tps_grid(t, dev, h, w, 25), lambda t: oracle.tps_grid(t, c), [(theta, True)], cuda,
                    atol=1e-5, grel=1e-3, rtol=1e-4, what="tps")  # fp32 sums over 49152 pixels, different order
    # the kernel evaluates the reference's formula in fp64 and rounds once: on the SAME theta it reproduces the fp64 oracle to
    # fp32 rounding (1 ulp of |grid| <= 1.4 is 1.2e-7) - the north star's "bit-exact for the TPS index grid" up to the final
    # rounding, and 40x inside SURVEY 0-6's 1e-5
    c64 = {k: (v.double() if torch.is_tensor(v) else v) for k, v in c.items()}
    exact = oracle.tps_grid(theta.double(), c64)
    ours = ops.tps_grid(theta.to(cuda), dev, h, w, 25).cpu().double()
    assert float((ours - exact).abs().max()) <= 2.5e-7, float((ours - exact).abs().max())
    # known answer: theta = 0 -> identity grid (base grid exactly, up to fp32 round-off of the affine part)
    g0 = ops.tps_grid(torch.zeros(1, 50, device=cuda), dev, h, w, 25).cpu()
    X, Y = c["gx"][None, None, :].expand(1, h, w), c["gy"][None, :, None].expand(1, h, w)
    assert_close(g0, torch.stack([X, Y], 3), atol=2e-5, what="tps identity")


@pytest.mark.parametrize("mode", ["border", "zeros"])
def test_grid_sample(ops, cuda, mode):
    inp = rnd(2, 3, 16, 12, seed=36)
    grid = (torch.rand(2, 16, 12, 2, generator=torch.Generator().manual_seed(37)) * 2.6 - 1.3)
    compare_fwd_bwd(lambda i, g: ops.grid_sample(i, g, mode),
                    lambda i, g: F.grid_sample(i, g, mode="bilinear", padding_mode=mode, align_corners=False),
                    [(inp, True), (grid, True)], cuda, atol=1e-5, gatol=1e-4, what=f"grid_sample {mode}")


@pytest.mark.parametrize("mode", ["border", "zeros"])
def test_grid_sample_taps_bit_exact(ops, cuda, mode):
    """The integer tap indices derived from the grid are bit-exact with ATen's CPU rule, at full size."""
    c, dev = _tps_consts(cuda, 256, 192, 5)
    theta = rnd(4, 50, seed=38, scale=0.2)
    grid = oracle.tps_grid(theta, c)  # same grid fed to both sides
    grid[0, 0, 0] = torch.tensor([-1.0, -1.0])
    grid[0, 0, 1] = torch.tensor([1.0, 1.0])
    grid[0, 0, 2] = torch.tensor([-1.0 + 1.0 / 192, 1.0 - 1.0 / 256])  # exactly on a pixel centre
    inp = rnd(4, 3, 256, 192, seed=39)
    out, taps = ops.grid_sample_taps(inp.to(cuda), grid.to(cuda), mode)
    assert torch.equal(taps.cpu(), oracle.grid_sample_taps(grid, 256, 192, mode))
    assert_close(out, F.grid_sample(inp, grid, mode="bilinear", padding_mode=mode, align_corners=False), atol=1e-5,
                 what="grid_sample full size")


def test_grid_sample_identity_known_answer(ops, cuda):
    """Sampling with the pixel-centre grid (align_corners=False) reproduces the input."""
    h, w = 16, 12
    xs = (torch.arange(w) * 2 + 1).float() / w - 1
    ys = (torch.arange(h) * 2 + 1).float() / h - 1
    grid = torch.stack([xs[None, :].expand(h, w), ys[:, None].expand(h, w)], 2)[None]
    inp = rnd(1, 3, h, w, seed=40)
    assert_close(ops.grid_sample(inp.to(cuda), grid.to(cuda), "border"), inp, atol=1e-6, what="identity sample")


def test_resample2d(ops, cuda):
    inp = rnd(2, 3, 16, 12, seed=41)
    flow = rnd(2, 2, 16, 12, seed=42, scale=2.0)
    compare_fwd_bwd(ops.resample2d, oracle.resample2d, [(inp, True), (flow, True)], cuda, atol=1e-5, gatol=1e-4,
                    what="resample2d")


# ------------------------------------------------------------------------------------------------ losses
def test_l1_loss(ops, cuda):
    a, b = rnd(2, 3, 32, 24, seed=43), rnd(2, 3, 32, 24, seed=44)
    compare_fwd_bwd(lambda p, q: ops.l1_loss(p, q), lambda p, q: F.l1_loss(p, q), [(a, True), (b, False)], cuda,
                    atol=1e-6, gatol=1e-9, what="l1 planar")
    compare_fwd_bwd(lambda p, q: ops.l1_loss(ops.activation(p, None), q, 0.25), lambda p, q: 0.25 * F.l1_loss(p, q),
                    [(a, True), (b, False)], cuda, atol=1e-6, gatol=1e-9, what="l1 rows")
    assert ops.l1_loss(a.to(cuda), a.to(cuda)).item() == 0.0


def test_scalar_sum_adds_left_to_right(ops, cuda):
    t = [torch.tensor(v, dtype=torch.float32, device=cuda, requires_grad=True) for v in (1e8, 1.0, -1e8, 3.0)]
    for k in (2, 3, 4):
        got = ops.scalar_sum(*t[:k])
        want = t[0].detach() + t[1].detach()
        for x in t[2:k]:
            want = want + x.detach()
        assert torch.equal(got.detach(), want), (k, got.item(), want.item())
    g = torch.autograd.grad(ops.scalar_sum(*t) * 2.0, t)
    assert all(float(x) == 2.0 for x in g)


def test_cat_channels_takes_planar_sources_straight_into_the_slab(ops, cuda):
    """get_and_cat_inputs on batch tensors: NCHW sources are transposed directly into their channel range."""
    a, b, c = rnd(2, 22, 32, 24, seed=63).to(cuda), rnd(2, 3, 32, 24, seed=64).to(cuda), rnd(2, 1, 32, 24, seed=65).to(cuda)
    rows = ops.activation(b, None)   # an NHWC-pitch tensor among planar ones
    out = ops.cat_channels([a, rows, c])
    assert torch.equal(out, torch.cat([a, b, c], 1))
    # 26 channels: the slab's pitch is 28 and the two pad columns are zero, so a convolution reads it in place
    assert ops._is_rows(out) and ops._ld(out) == 28 and getattr(out, "_so_zero_padded", 0) == 28
    padded = torch.as_strided(out, (2, 28, 32, 24), out.stride(), out.storage_offset())
    assert float(padded[:, 26:].abs().max()) == 0.0
    w = rnd(8, 26, 3, 3, seed=66, scale=0.1).to(cuda)
    assert_close(ops.conv2d(out, w, None, 1, 1), F.conv2d(torch.cat([a, b, c], 1).cpu(), w.cpu(), None, 1, 1), atol=2e-5,
                 what="conv on the zero-padded concatenation")
    out4 = ops.cat_channels([a, a[:, :2].contiguous()])   # 24 channels: dense
    assert ops._ld(out4) == 24 and out4.permute(0, 2, 3, 1).is_contiguous() and torch.equal(out4, torch.cat([a, a[:, :2]], 1))


def test_tryon_compose_and_blend(ops, cuda):
    o, cloth = rnd(2, 4, 16, 12, seed=45), rnd(2, 3, 16, 12, seed=46)

    def ref(o_, c_):
        r, m = torch.tanh(o_[:, :3]), torch.sigmoid(o_[:, 3:4])
        return r, m, (1 - m) * r + m * c_

    compare_fwd_bwd(ops.tryon_compose, ref, [(o, True), (cloth, False)], cuda, atol=1e-6, gatol=1e-5, what="compose")
    a, b, m = rnd(2, 3, 8, 6, seed=47), rnd(2, 3, 8, 6, seed=48), torch.sigmoid(rnd(2, 1, 8, 6, seed=49))
    compare_fwd_bwd(ops.blend, lambda a_, b_, m_: (1 - m_) * a_ + m_ * b_, [(a, True), (b, True), (m, True)], cuda,
                    atol=1e-6, gatol=1e-5, what="blend")
    compare_fwd_bwd(ops.tensor_sum, lambda t: t.sum(), [(m, True)], cuda, atol=1e-4, what="sum")


def test_adam_matches_torch(ops, cuda):
    p0, steps = rnd(1000, seed=50), 5
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-2)
    p, m, v = p0.clone().to(cuda), torch.zeros(1000, device=cuda), torch.zeros(1000, device=cuda)
    for s in range(1, steps + 1):
        g = rnd(1000, seed=60 + s)
        p_ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.to(cuda), m, v, 1e-2, 0.9, 0.999, 1e-8, s)
    assert_close(p, p_ref.detach(), atol=1e-6, what="adam")


@pytest.mark.parametrize("M,N,K,batch", [(1536, 256, 256, 36), (384, 512, 512, 36), (3072, 128, 512, 16), (1000, 132, 128, 16),
                                         (770, 128, 192, 36), (96 * 12, 256, 1024, 16)])
def test_persistent_winograd_gemm_is_bit_identical_to_the_engine(cuda, M, N, K, batch):
    """csrc/pgemm.hip (persistent workgroups, LDS-DMA pipeline kept running across output tiles) against the general engine's
    batched NT GEMM on Winograd-domain shapes, ragged M / N included: same tile, same k order, same MFMA sequence ->
    torch.equal.  Run twice into a dirtied output (every element must be rewritten)."""
    from shineon_virtual_tryon_amd import lib

    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(5)
    A = torch.randn(batch, M, K, generator=g).to(cuda)
    B = torch.randn(batch, N, K, generator=g).to(cuda)
    ws = torch.empty(64 << 20, device=cuda)
    ref = torch.empty(batch, M, N, device=cuda)
    L.so_igemm_force(64, 64, 1)
    try:
        assert L.so_gemm_batched(0, 1, M, N, K, A.data_ptr(), K, M * K, B.data_ptr(), K, N * K, ref.data_ptr(), N, M * N, batch, None, None,
                                 None, 0, 0, 0, 0.0, ws.data_ptr(), ws.numel() * 4, st) == 0
    finally:
        L.so_igemm_force(0, 0, 0)
    for rep in range(2):
        out = torch.full((batch, M, N), float("nan"), device=cuda)
        rc = L.so_pgemm_nt(M, N, K, A.data_ptr(), K, M * K, B.data_ptr(), K, N * K, out.data_ptr(), N, M * N, batch, st)
        assert rc == 0, rc   # 0 = launched (SO_NOT_APPLICABLE = -3 would mean it declined)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), (rep, float((out - ref).abs().max()))
    # and against fp64 on a slice
    want = A[3].double().cpu() @ B[3].double().cpu().T
    assert (ref[3].double().cpu() - want).abs().max().item() <= 2e-4 * want.abs().max().item()


# ------------------------------------------------------------------------------------------------ Winograd F(2x2, 3x3)
@pytest.mark.parametrize("n,h,w,c,ko,act,gated", [
    (2, 16, 12, 64, 32, 1, False),     # ReLU epilogue (VGG forward)
    (1, 7, 5, 8, 12, 0, False),        # odd H and W: ragged last tile row / column
    (3, 8, 6, 128, 128, 0, True),      # gate = ReLU backward fused (VGG input gradient)
    (8, 32, 24, 256, 512, 1, False),   # VGG conv4_1 at 2B = 8 images
])
def test_winograd_conv3x3_matches_fp64_convolution(cuda, ops, n, h, w, c, ko, act, gated):
    """csrc/wino.hip against an fp64 convolution of the same operands, next to the direct implicit-GEMM kernel: the
    Winograd result must be as close to the exact value as the direct fp32 sum is (within 4x its error + 2e-6 of max)."""
    from shineon_virtual_tryon_amd import lib

    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    x = rnd(n, h, w, c, seed=1)
    wt = rnd(ko, 3, 3, c, seed=2, scale=(2.0 / (9 * c)) ** 0.5)
    bias = rnd(ko, seed=3, scale=0.1)
    gate = rnd(n, h, w, ko, seed=4) if gated else None
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.permute(0, 3, 1, 2).double(), bias.double(), padding=1)
    ref = (ref.relu() if act else ref).permute(0, 2, 3, 1)
    if gated:
        ref = torch.where(gate.double() > 0, ref, torch.zeros_like(ref))
    xd, wd, bd = x.to(cuda), wt.to(cuda), bias.to(cuda)
    gd = gate.to(cuda) if gated else None
    u = torch.empty(16, ko, c, device=cuda)
    assert L.so_wino_weights(wd.data_ptr(), u.data_ptr(), ko, ko, c, 0, st) == 0
    wws = torch.empty(L.so_wino_ws_floats(n, h, w, c, ko), device=cuda)
    ws = ops.workspace(cuda)
    y = torch.full((n, h, w, ko), float("nan"), device=cuda)
    assert L.so_wino_conv3x3(xd.data_ptr(), c, u.data_ptr(), bd.data_ptr(), ko, gd.data_ptr() if gated else None, y.data_ptr(), ko,
                             n, h, w, c, ko, act, 0.0, wws.data_ptr(), wws.numel() * 4, ws.data_ptr(), ws.numel() * 4, st) == 0
    yd = torch.empty_like(y)
    assert L.so_conv2d_fprop(xd.data_ptr(), c, wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), ko, n, h, w, c, ko, 3, 3, 1, 1, act, 0.0,
                             ws.data_ptr(), ws.numel() * 4, st) == 0
    if gated:
        yd = torch.where(gd > 0, yd, torch.zeros_like(yd))
    big = float(ref.abs().max())
    e_w, e_d = float((y.cpu().double() - ref).abs().max()), float((yd.cpu().double() - ref).abs().max())
    print(f"winograd {n}x{h}x{w} {c}->{ko}: max err {e_w:.2e} (direct {e_d:.2e}), max |y| {big:.2f}")
    assert e_w <= 4 * e_d + 2e-6 * big, (e_w, e_d, big)
    # the fused single-launch form (weights in its own [c/8][xi][ko][8] order)
    uf = torch.empty(L.so_wino_fused_weight_floats(ko, c, 0), device=cuda)
    assert L.so_wino_fused_weights(wd.data_ptr(), uf.data_ptr(), ko, ko, c, 0, st) == 0
    y2 = torch.full((n, h, w, ko), float("nan"), device=cuda)
    assert L.so_wino_fused_conv3x3(xd.data_ptr(), c, uf.data_ptr(), bd.data_ptr(), ko, gd.data_ptr() if gated else None, y2.data_ptr(),
                                   ko, n, h, w, c, ko, act, 0.0, st) == 0
    e_f = float((y2.cpu().double() - ref).abs().max())
    print(f"   fused: max err {e_f:.2e}")
    assert e_f <= 4 * e_d + 2e-6 * big, (e_f, e_d, big)
    if not gated and h % 2 == 0 and w % 2 == 0:
        # ... with MaxPool2d(2, 2) written by the epilogue: the pooled map is bit-identical to pooling y2, y itself is stored
        # for the first n_keep images only (the rest of the buffer is never touched)
        for n_keep in sorted({n, n // 2, 0}):
            y3 = torch.full((n, h, w, ko), float("nan"), device=cuda)
            yp = torch.full((n, h // 2, w // 2, ko), float("nan"), device=cuda)
            assert L.so_wino_fused_conv3x3_pool(xd.data_ptr(), c, uf.data_ptr(), bd.data_ptr(), ko, y3.data_ptr(), ko, n_keep,
                                                yp.data_ptr(), ko, n, h, w, c, ko, act, 0.0, st) == 0
            want = F.max_pool2d(y2.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
            assert torch.equal(yp, want), ("pooled epilogue", n_keep)
            assert torch.equal(y3[:n_keep], y2[:n_keep]) and bool(torch.isnan(y3[n_keep:]).all()), ("n_keep", n_keep)
        assert L.so_wino_fused_conv3x3_pool(xd.data_ptr(), c, uf.data_ptr(), bd.data_ptr(), ko, y3.data_ptr(), ko, n, None, ko, n, h, w,
                                            c, ko, act, 0.0, st) != 0, "a missing pooled output must be refused"
    # F(4x4, 3x3) (36 transform points, the deep layers): its fp32 round-off is of the order of the direct K = 9C MFMA chain's
    u4 = torch.empty(36, ko, c, device=cuda)
    assert L.so_wino4_weights(wd.data_ptr(), u4.data_ptr(), ko, ko, c, 0, st) == 0
    wws4 = torch.empty(L.so_wino4_ws_floats(n, h, w, c, ko), device=cuda)
    y4 = torch.full((n, h, w, ko), float("nan"), device=cuda)
    assert L.so_wino4_conv3x3(xd.data_ptr(), c, u4.data_ptr(), bd.data_ptr(), ko, gd.data_ptr() if gated else None, y4.data_ptr(), ko,
                              n, h, w, c, ko, act, 0.0, wws4.data_ptr(), wws4.numel() * 4, ws.data_ptr(), ws.numel() * 4, st) == 0
    e_4 = float((y4.cpu().double() - ref).abs().max())
    print(f"   F(4x4,3x3): max err {e_4:.2e}")
    assert e_4 <= 4 * e_d + 1.2e-5 * big, (e_4, e_d, big)


def test_winograd_input_gradient_is_the_flipped_transposed_convolution(cuda, ops):
    """dx of a 3x3 / s1 / p1 convolution through so_wino_weights(flip_transpose=1) + so_wino_conv3x3 against autograd (fp64)."""
    from shineon_virtual_tryon_amd import lib

    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    n, h, w, c, ko = 2, 16, 12, 64, 128
    x = rnd(n, c, h, w, seed=5).double().requires_grad_(True)
    wt = rnd(ko, 3, 3, c, seed=6, scale=0.05)
    dy = rnd(n, h, w, ko, seed=7)
    F.conv2d(x, wt.permute(0, 3, 1, 2).double(), padding=1).backward(dy.permute(0, 3, 1, 2).double())
    ref = x.grad.permute(0, 2, 3, 1)
    wd, dyd = wt.to(cuda), dy.to(cuda)
    u = torch.empty(16, c, ko, device=cuda)
    assert L.so_wino_weights(wd.data_ptr(), u.data_ptr(), ko, ko, c, 1, st) == 0
    wws = torch.empty(L.so_wino_ws_floats(n, h, w, ko, c), device=cuda)
    ws = ops.workspace(cuda)
    dx = torch.empty(n, h, w, c, device=cuda)
    assert L.so_wino_conv3x3(dyd.data_ptr(), ko, u.data_ptr(), None, 0, None, dx.data_ptr(), c, n, h, w, ko, c, 0, 0.0,
                             wws.data_ptr(), wws.numel() * 4, ws.data_ptr(), ws.numel() * 4, st) == 0
    assert_close(dx, ref.float(), atol=3e-6 * float(ref.abs().max()), what="winograd input gradient")
    uf = torch.empty(L.so_wino_fused_weight_floats(ko, c, 1), device=cuda)
    assert L.so_wino_fused_weights(wd.data_ptr(), uf.data_ptr(), ko, ko, c, 1, st) == 0
    dx2 = torch.full((n, h, w, c), float("nan"), device=cuda)
    assert L.so_wino_fused_conv3x3(dyd.data_ptr(), ko, uf.data_ptr(), None, 0, None, dx2.data_ptr(), c, n, h, w, ko, c, 0, 0.0, st) == 0
    assert_close(dx2, ref.float(), atol=3e-6 * float(ref.abs().max()), what="fused winograd input gradient")
    u4 = torch.empty(36, c, ko, device=cuda)
    assert L.so_wino4_weights(wd.data_ptr(), u4.data_ptr(), ko, ko, c, 1, st) == 0
    wws4 = torch.empty(L.so_wino4_ws_floats(n, h, w, ko, c), device=cuda)
    dx4 = torch.full((n, h, w, c), float("nan"), device=cuda)
    assert L.so_wino4_conv3x3(dyd.data_ptr(), ko, u4.data_ptr(), None, 0, None, dx4.data_ptr(), c, n, h, w, ko, c, 0, 0.0,
                              wws4.data_ptr(), wws4.numel() * 4, ws.data_ptr(), ws.numel() * 4, st) == 0
    assert_close(dx4, ref.float(), atol=1.5e-5 * float(ref.abs().max()), what="F(4x4,3x3) input gradient")


def test_bounded_bucket_wait_returns_on_abort_and_on_deadline(cuda):
    """The polling wait of the overlapped gradient exchange cannot spin forever: queued behind a signal
    word that nobody ever stores, it returns when the host sets the abort word (status 1) or when its 100 MHz wall-clock
    deadline passes (status 2), and a word that IS signalled releases it with status 0."""
    import ctypes
    import time

    from shineon_virtual_tryon_amd import lib

    L = lib()
    flag = L.so_signal_alloc()
    assert flag
    counter = torch.zeros(2, dtype=torch.int32, device=cuda)
    comm = torch.cuda.Stream()
    try:
        # 1. deadline: 0.05 s, nobody signals
        words = L.so_hostwords_alloc()
        w = (ctypes.c_uint32 * 2).from_address(words)
        assert w[0] == 0 and w[1] == 0
        t0 = time.time()
        assert L.so_stream_wait_ge_bounded(flag, 1, words, int(0.05 * 1e8), comm.cuda_stream) == 0
        comm.synchronize()
        dt = time.time() - t0
        assert w[1] == 2 and 0.04 <= dt < 5.0, (w[1], dt)
        L.so_hostwords_free(words)
        # 2. abort: a 30 s deadline, released by the host-side abort word within milliseconds
        words = L.so_hostwords_alloc()
        w = (ctypes.c_uint32 * 2).from_address(words)
        assert L.so_stream_wait_ge_bounded(flag, 1, words, int(30 * 1e8), comm.cuda_stream) == 0
        time.sleep(0.05)
        assert not comm.query()          # still polling
        w[0] = 1
        t0 = time.time()
        comm.synchronize()
        assert w[1] == 1 and time.time() - t0 < 2.0, (w[1], time.time() - t0)
        L.so_hostwords_free(words)
        # 3. the normal case: counter bumped to 1 and stored into the word by a kernel of another stream
        words = L.so_hostwords_alloc()
        w = (ctypes.c_uint32 * 2).from_address(words)
        assert L.so_stream_wait_ge_bounded(flag, 1, words, int(30 * 1e8), comm.cuda_stream) == 0
        st = torch.cuda.current_stream().cuda_stream
        assert L.so_counter_bump(counter.data_ptr(), st) == 0
        assert L.so_signal_store(flag, counter.data_ptr(), 0, st) == 0
        comm.synchronize()
        assert w[1] == 0 and int(counter[0]) == 1
        L.so_hostwords_free(words)
    finally:
        L.so_signal_free(flag)

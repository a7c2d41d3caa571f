"""GPU parity of the SAMS-GAN path through the C ABI: the SAMS-only kernels against plain PyTorch
fp32 references of the same ops, and SamsModel's three training steps against oracle/sams_oracle.py (itself pinned to the
reference's SamsModel by tests/golden/sams_*.npz) — every logged scalar, the generated frames, every gradient of each
step's parameter set element-wise, and the buffers the steps mutate."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sams_helpers as sh
from oracle import sams_oracle as so
from oracle.procedural import procedural_state_dict

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(autouse=True)
def fixture_cases_run_on_committed_plans(request):
    """The cases compared with committed fixtures (three training steps of the small variants, the full-size steps and
    generator pass) launch only layer shapes with a COMMITTED igemm plan (tools/gpu_make_plans.sh collects them from these
    very tests): the kernel choice, hence the split-K order, hence every bit is the same on every box, which lets
    tests/golden/full/routes.json pin their acceptance routes strictly (round 4: measured per process, route changes were
    warnings)."""
    import os

    import shineon_virtual_tryon_amd as pkg

    pinned = any(t in request.node.name for t in ("three_training_steps", "full_size_generator_pass", "two_full_iterations"))
    L = pkg.lib()
    before = L.so_igemm_plan_count()
    yield
    if pinned and not os.environ.get("SHINEON_PLANS_SAVE"):
        assert L.so_igemm_plan_count() == before, (f"{L.so_igemm_plan_count() - before} layer shapes without a committed plan "
                                                    "were measured: regenerate plans/gfx950.txt (tools/gpu_make_plans.sh)")


def _ops():
    from shineon_virtual_tryon_amd import ops, ops_sams

    return ops, ops_sams


def ops_to_oihw(t):
    """A parameter gradient in the logical (O, I, R, S) element order (the memory is OHWI)."""
    return t.detach().contiguous()


def _nchw(t):
    ops, _ = _ops()
    return ops.to_nchw(t).cpu()


# ------------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,size,scale", [
    ((2, 8, 16, 12), (8, 6), None), ((2, 8, 16, 12), (4, 3), None), ((3, 5, 64, 48), (8, 6), None),
    ((2, 8, 8, 6), None, 2), ((2, 6, 16, 12), None, 0.5), ((1, 4, 7, 5), (3, 2), None), ((1, 3, 5, 4), (13, 9), None),
    ((2, 4, 256, 192), (16, 12), None),
])
def test_resize_nearest_bit_exact_forward_and_adjoint(shape, size, scale):
    ops, ops_sams = _ops()
    torch.manual_seed(1)
    x = torch.randn(shape)
    ref_in = x.clone().requires_grad_(True)
    ref = F.interpolate(ref_in, size=size, scale_factor=scale, mode="nearest")
    g = torch.randn_like(ref)
    ref.backward(g)
    xin = x.to(DEV).requires_grad_(True)
    y = ops_sams.resize_nearest(xin, size=size, scale_factor=scale)
    y.backward(g.to(DEV))
    assert torch.equal(_nchw(y), ref.detach())  # a gather: bit-exact
    # the adjoint sums at most a few terms per pixel; same terms, possibly another order
    assert torch.allclose(_nchw(xin.grad), ref_in.grad, atol=1e-6, rtol=0)


@pytest.mark.parametrize("shape", [(2, 15, 64, 48), (4, 3, 9, 7), (1, 6, 256, 192), (2, 4, 1, 5)])
def test_avg_pool_3x3_stride_2_no_pad_count(shape):
    ops, ops_sams = _ops()
    torch.manual_seed(2)
    x = torch.randn(shape)
    ref_in = x.clone().requires_grad_(True)
    ref = F.avg_pool2d(ref_in, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)
    g = torch.randn_like(ref)
    ref.backward(g)
    xin = x.to(DEV).requires_grad_(True)
    y = ops_sams.avg_pool3s2(xin)
    y.backward(g.to(DEV))
    assert y.shape == ref.shape
    assert torch.allclose(_nchw(y), ref.detach(), atol=1e-6, rtol=0)
    assert torch.allclose(_nchw(xin.grad), ref_in.grad, atol=1e-6, rtol=0)


@pytest.mark.parametrize("act", ["none", "leaky", "gelu", "swish"])
@pytest.mark.parametrize("c", [8, 6])
def test_spade_modulation_forward_backward(act, c):
    ops, ops_sams = _ops()
    torch.manual_seed(3)
    n = torch.randn(2, c, 9, 7)
    gb = torch.randn(2, 2 * c, 9, 7) * 0.5
    g = torch.randn(2, c, 9, 7)

    def ref_act(t):
        return {"none": t, "leaky": F.leaky_relu(t, 0.2), "gelu": F.gelu(t), "swish": t * torch.sigmoid(t)}[act]

    rn, rgb = n.clone().requires_grad_(True), gb.clone().requires_grad_(True)
    ref = ref_act(rn * (1 + rgb[:, :c]) + rgb[:, c:])
    ref.backward(g)
    dn, dgb = n.to(DEV).requires_grad_(True), gb.to(DEV).requires_grad_(True)
    y = ops_sams.spade_modulate(dn, dgb, act, 0.2 if act == "leaky" else 0.0)
    y.backward(g.to(DEV))
    assert torch.allclose(_nchw(y), ref.detach(), atol=2e-6, rtol=1e-6)
    assert torch.allclose(_nchw(dn.grad), rn.grad, atol=2e-6, rtol=1e-5)
    assert torch.allclose(_nchw(dgb.grad), rgb.grad, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("shape", [(16, 8, 3, 3), (32, 16, 1, 1), (64, 32, 4, 4), (1024, 1024, 3, 3), (6, 5, 3, 3)])
def test_spectral_norm_power_iteration_weight_and_gradient(shape):
    ops, ops_sams = _ops()
    torch.manual_seed(4)
    o = shape[0]
    k = int(np.prod(shape[1:]))
    w = torch.randn(shape) * 0.1
    u = F.normalize(torch.randn(o), dim=0)
    v = F.normalize(torch.randn(k), dim=0)
    g = torch.randn(shape)
    for training in (True, False):
        sd = {"c.weight_orig": w.clone().double().requires_grad_(True), "c.weight_u": u.clone().double(),
              "c.weight_v": v.clone().double()}
        ref = so.spectral_weight(sd, "c", training)
        ref.backward(g.double())
        wd = torch.empty(shape[0], shape[2], shape[3], shape[1], device=DEV).permute(0, 3, 1, 2)  # OHWI memory, like HipConv2d
        wd.copy_(w)
        wd.requires_grad_(True)
        ud, vd = u.to(DEV), v.to(DEV)
        out = ops_sams.spectral_normalize(wd, ud, vd, training)
        out.backward(g.to(DEV))
        scale = ref.detach().abs().max().item()
        assert torch.allclose(out.detach().cpu().double(), ref.detach(), atol=3e-6 * scale, rtol=0), (shape, training)
        assert torch.allclose(ud.cpu().double(), sd["c.weight_u"], atol=5e-6), (shape, training)
        assert torch.allclose(vd.cpu().double(), sd["c.weight_v"], atol=5e-6), (shape, training)
        gref = sd["c.weight_orig"].grad
        assert torch.allclose(wd.grad.cpu().double(), gref, atol=2e-5 * gref.abs().max().item(), rtol=0), (shape, training)
        if not training:
            assert torch.equal(ud.cpu(), u) and torch.equal(vd.cpu(), v)


@pytest.mark.parametrize("mode", ["hinge", "ls", "original", "w"])
@pytest.mark.parametrize("real,for_disc", [(True, True), (False, True), (True, False)])
def test_gan_losses(mode, real, for_disc):
    ops, ops_sams = _ops()
    torch.manual_seed(5)
    x = torch.randn(4, 1, 9, 7) * 2
    x[0, 0, 0, :3] = torch.tensor([1.0, -1.0, 0.0])  # the hinge's kinks: torch.min splits the gradient on a tie
    rx = x.clone().requires_grad_(True)
    ref = so.gan_loss_single(rx, mode, real, for_disc)
    ref.backward()
    dx = x.to(DEV).requires_grad_(True)
    out = ops_sams.gan_loss(dx, mode, real, for_disc)
    out.backward()
    assert out.shape == ()
    assert abs(out.item() - ref.item()) <= 1e-6 * max(1.0, abs(ref.item()))
    assert torch.allclose(_nchw(dx.grad), rx.grad, atol=1e-8, rtol=1e-5)


def test_gan_loss_hinge_generator_must_aim_for_real():
    ops, ops_sams = _ops()
    with pytest.raises(AssertionError):
        ops_sams.gan_loss(torch.zeros(1, 1, 2, 2, device=DEV), "hinge", False, False)


def _compare_grads(got, ref32, ref64, what, kink=None):
    """Rule of the try-on path (tests/test_parity_bs4_gpu.py) — within 2e-3 * max of the fp32 oracle OR of the fp64 oracle;
    analytically-zero gradients (fp64 says 0) stay at the oracle's own noise level — plus a conditioning term: the fp32
    oracle's own distance from the fp64 one (`own`) measures how much THIS tensor moves under rounding.  It is ~1e-6 * max
    for almost every tensor; 5 * own is granted (10 * own for scalars, whose cancellation makes them the worst conditioned).
    kink: {key: spread} from sams_helpers.kink_spread — how far the fp64 gradient moves when a pre-activation within 1e-5
    of a ReLU / LeakyReLU kink takes the other side (0 unless such an element exists; one pixel of these 64 x 48 frames
    then moves a gradient by up to 10 %, in any fp32 implementation)."""
    assert set(got) == set(ref32), (what, set(got) ^ set(ref32))
    for k in sorted(got):
        g, a, b = got[k].double().cpu(), ref32[k].double(), ref64[k].double()
        big = max(a.abs().max().item(), b.abs().max().item())
        if b.abs().max().item() <= 1e-6 * max(a.abs().max().item(), 1e-30):
            noise = max((a - b).abs().max().item(), 1e-12)
            assert (g - b).abs().max().item() <= 10 * noise + 1e-10, (what, k, "analytic zero")
            continue
        own = (a - b).abs().max().item()
        err = min((g - a).abs().max().item(), (g - b).abs().max().item())
        tol = 2e-3 * big + (10 if g.numel() == 1 else 5) * own
        if err > tol and kink is not None:
            if callable(kink):  # two more fp64 oracle runs: only made when a tensor needs them
                kink = kink()
            per_step = kink[int(what.rsplit(" ", 1)[1])] if isinstance(kink, list) else kink
            tol += 1.5 * per_step.get(k, 0.0)
        assert err <= tol, (what, k, err, tol, big)
    return kink


@pytest.mark.parametrize("norm_G", ["spectralspadesyncbatch3x3", "spadeinstance3x3", "spadebatch3x3"])
@pytest.mark.parametrize("fin,fout,hw", [(32, 32, (16, 12)), (32, 16, (16, 12)), (16, 32, (32, 24)), (8, 8, (4, 3))])
@pytest.mark.parametrize("multi", [False, True])
def test_spade_residual_block_against_the_oracle(norm_G, fin, fout, hw, multi):
    """One AnySpadeResBlock (plain SPADE as in the encoder, MultiSpade as in the middle / decoder), forward and every
    gradient, against oracle.spade_resblock on the same procedural weights."""
    import argparse

    from oracle.procedural import shapes_of
    from shineon_virtual_tryon_amd.networks.sams import SPADE, AnySpadeResBlock, MultiSpade

    labels = {"agnostic": 4, "cloth": 3, "flow": 2} if multi else 8
    block = AnySpadeResBlock(fin, fout, norm_G, labels, MultiSpade if multi else SPADE, "relu")
    sd = procedural_state_dict({"b." + k: v for k, v in shapes_of(block.state_dict()).items()})
    block.load_state_dict({k[2:]: v for k, v in sd.items()})
    block = block.to(DEV).train()
    torch.manual_seed(7)
    b, (h, w) = 2, hw
    x = torch.randn(b, fin, h, w)
    seg = {k: torch.randn(b, c, 64, 48) for k, c in labels.items()} if multi else torch.randn(b, 8, 64, 48)
    gout = torch.randn(b, fout, h, w)
    hp = argparse.Namespace(norm_G=norm_G, activation="relu")
    refs = []
    for dtype in (torch.float32, torch.float64):
        osd = {k: (v.to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for k, v in osd.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var", "weight_u", "weight_v")):
                v.requires_grad_(True)
        xin = x.clone().to(dtype).requires_grad_(True)
        oseg = {k: v.to(dtype) for k, v in seg.items()} if multi else seg.to(dtype)
        out = so.spade_resblock(osd, "b", xin, oseg, hp, True)
        out.backward(gout.to(dtype))
        refs.append((out.detach(), xin.grad, {k[2:]: v.grad for k, v in osd.items() if v.requires_grad and v.grad is not None}))
    dx = x.clone().to(DEV).requires_grad_(True)
    dseg = {k: v.to(DEV) for k, v in seg.items()} if multi else seg.to(DEV)
    y = block(dx, dseg)
    y.backward(gout.to(DEV))
    (o32, gx32, g32), (o64, gx64, g64) = refs
    big = o64.abs().max().item()
    assert (_nchw(y).double() - o64).abs().max().item() <= 1e-4 * big
    _compare_grads({"x": dx.grad, **{k: p.grad for k, p in block.named_parameters() if p.grad is not None}},
                   {"x": gx32, **g32}, {"x": gx64, **g64}, f"{norm_G} {fin}->{fout} {hw}")


@pytest.mark.parametrize("norm_D,num_D,n_layers,keep_feats", [
    ("spectralinstance", 2, 4, True), ("spectralbatch", 3, 3, True), ("spectralsync_batch", 1, 4, False),
    ("spectralnone", 2, 5, True), ("spectral", 2, 2, False)])
def test_patchgan_discriminators_against_the_oracle(norm_D, num_D, n_layers, keep_feats):
    """MultiscaleDiscriminator (and through it NLayerDiscriminator) for every norm_D the reference accepts: all stage
    outputs of all scales, the gradient of a weighted sum of them with respect to the input and every parameter, and
    the buffers (u, v, BatchNorm statistics and counters) after the pass."""
    import argparse

    from oracle.procedural import shapes_of
    from shineon_virtual_tryon_amd.networks.discriminator import MultiscaleDiscriminator

    hp = sh.sams_hparams(norm_D=norm_D, num_D=num_D, n_layers_D=n_layers, ndf=8, no_ganFeat_loss=not keep_feats)
    net = MultiscaleDiscriminator(hp)
    sd = procedural_state_dict({"d." + k: v for k, v in shapes_of(net.state_dict()).items()})
    net.load_state_dict({k[2:]: v for k, v in sd.items()})
    net = net.to(DEV).train()
    torch.manual_seed(8)
    x = torch.randn(4, 15, 64, 48)
    refs = []
    for dtype in (torch.float32, torch.float64):
        osd = {k: (v.to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for k, v in osd.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var", "weight_u", "weight_v")):
                v.requires_grad_(True)
        xin = x.clone().to(dtype).requires_grad_(True)
        outs = [t for scale in so.multiscale_discriminator(osd, "d", xin, hp, True) for t in scale]
        torch.manual_seed(9)
        ws = [torch.randn(t.shape) for t in outs]
        sum((t * w.to(dtype)).sum() for t, w in zip(outs, ws)).backward()
        refs.append(([t.detach() for t in outs], xin.grad, {k[2:]: v.grad for k, v in osd.items() if v.requires_grad and v.grad is not None}, osd))
    dx = x.clone().to(DEV).requires_grad_(True)
    got = [t for scale in net(dx) for t in scale]
    assert len(got) == len(refs[0][0]) == num_D * ((n_layers + 1) if keep_feats else 1)
    from shineon_virtual_tryon_amd import ops

    sum(ops.tensor_sum(t * ops.to_rows(w.to(DEV))) for t, w in zip(got, ws)).backward()
    (o32, gx32, g32, _), (o64, gx64, g64, sd64) = refs
    for a, b in zip(got, o64):
        assert (_nchw(a).double() - b).abs().max().item() <= 1e-4 * max(b.abs().max().item(), 1e-6)
    _compare_grads({"x": dx.grad, **{k: p.grad for k, p in net.named_parameters() if p.grad is not None}},
                   {"x": gx32, **g32}, {"x": gx64, **g64}, f"D {norm_D}")
    for k, v in net.state_dict().items():
        ref = sd64["d." + k]
        if v.is_floating_point():
            if k.endswith(("weight_u", "weight_v", "running_mean", "running_var")):
                assert (v.cpu().double() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), k
        else:
            assert int(v) == int(ref), k


@pytest.mark.parametrize("kw", [
    dict(ngf_pow_outer=3, ngf_pow_inner=6, ngf_pow_step=2, num_middle=1),                      # overshoot: one extra block per side
    dict(norm_G="spadeinstance5x5", activation="swish"),                                       # 5x5 SPADE convolutions
    # (not "sine": sin(30 x) stacked 20 deep turns fp32 round-off into O(1) differences in ANY two implementations)
    dict(norm_G="spectralspadebatch3x3", activation="swish", attention_decoder_indices=["-1"]),  # attention, last decoder block
    dict(n_frames_total=2, flow_warp=False, activation="gelu"),                                # one previous frame, 3 output channels
], ids=["pow_step2", "instance5x5_swish", "decoder_attention_swish", "two_frames_gelu"])
def test_sams_generator_variants_against_the_oracle(kw):
    """SamsGenerator.forward for option combinations the three golden option sets do not reach: output, the gradient with
    respect to every parameter, and the buffers after the pass, against oracle.generator_forward."""
    from oracle.procedural import shapes_of
    from shineon_virtual_tryon_amd.networks.sams.sams_generator import SamsGenerator

    hp = sh.sams_hparams(**kw)
    gen = SamsGenerator(hp)
    sd = procedural_state_dict({"generator." + k: v for k, v in shapes_of(gen.state_dict()).items()})
    gen.load_state_dict({k[len("generator."):]: v for k, v in sd.items()})
    gen = gen.to(DEV).train()
    torch.manual_seed(12)
    b, n, h, w = 2, hp.n_frames_total, hp.fine_height, hp.fine_width
    prev_frames = torch.randn(b, n - 1, 3, h, w) * 0.5
    prev_maps = torch.randn(b, n - 1, 2, h, w)
    maps = {"agnostic": torch.randn(b, 4, h, w), "densepose": torch.randn(b, 3, h, w), "flow": torch.randn(b, 2, h, w),
            "cloth": torch.randn(b, 3, h, w)}
    gout = torch.randn(b, 4 if hp.flow_warp else 3, h, w)
    refs = []
    for dtype in (torch.float32, torch.float64):
        osd = {k: (v.to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for k, v in osd.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var", "weight_u", "weight_v")):
                v.requires_grad_(True)
        out = so.generator_forward(osd, prev_frames.to(dtype), prev_maps.to(dtype), {k: v.to(dtype) for k, v in maps.items()}, hp, True)
        out.backward(gout.to(dtype))
        refs.append((out.detach(), {k[len("generator."):]: v.grad for k, v in osd.items() if v.requires_grad and v.grad is not None}, osd))
    y = gen(prev_frames.to(DEV), prev_maps.to(DEV), {k: v.to(DEV) for k, v in maps.items()})
    y.backward(gout.to(DEV))
    (o32, g32, _), (o64, g64, sd64) = refs
    assert (_nchw(y).double() - o64).abs().max().item() <= 1e-4 * o64.abs().max().item()
    _compare_grads({k: p.grad for k, p in gen.named_parameters() if p.grad is not None}, g32, g64, f"generator {kw}")
    for k, v in gen.state_dict().items():
        if k.endswith(("weight_u", "weight_v", "running_mean", "running_var")):
            ref = sd64["generator." + k]
            assert (v.cpu().double() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), k


# ------------------------------------------------------------------------------------------------
# model
# ------------------------------------------------------------------------------------------------
def _model(tag, sd_cpu):
    from shineon_virtual_tryon_amd.sams_model import SamsModel

    hp = sh.sams_hparams(**sh.SAMS_VARIANTS[tag])
    hp.allow_random_vgg = True
    model = SamsModel(hp)
    model.load_state_dict(sd_cpu, strict=True)
    return model.to(DEV).train(), hp


def _batch(hp, bs=2):
    from shineon_virtual_tryon_amd.data import synthetic_batch

    return synthetic_batch(bs, "cpu", height=hp.fine_height, width=hp.fine_width, n_frames=hp.n_frames_total, smooth=True)


def _to(batch, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}


def _run_three_steps_against_fixture(model, dbatch, fix, what, golden=None, tag=None):
    """The three optimizer steps in Lightning's order on the HIP path; every logged scalar, every gradient of each step's
    parameter set (gradfix rule "sams": the r03 _compare_grads rule on every N-th element + whole-tensor energy, with the
    fp32 / fp64 / kink-bracket oracle values committed under tests/golden/full/), the generated frames."""
    import fullsize_cases as fc  # noqa: F401
    import gradfix as gf
    from shineon_virtual_tryon_amd.trainer import MultiOptimizerStep

    nets = model.optimizer_networks()
    stepper = MultiOptimizerStep.__new__(MultiOptimizerStep)  # only its requires_grad toggling is used here
    stepper._all = list(model.parameters())
    stepper._own = [list(n.parameters()) for n in nets]
    for idx, name in enumerate(sh.STEP_NETS):
        stepper._only(idx)
        model.zero_grad(set_to_none=True)
        res = model.training_step(dbatch, 0, idx)
        res.minimize.sum().backward()
        for k, v in res.logs.items():
            r32, r64 = float(fix[f"log32:{idx}:{k}"]), float(fix[f"log64:{idx}:{k}"])
            if golden is not None:  # the reference's own value
                gold = float(golden[f"log{idx}:{k}"])
                assert abs(float(v) - gold) <= 2e-4 * max(1.0, abs(gold)), (what, idx, k, float(v), gold)
                assert abs(float(v) - r64) <= 2e-4 * max(1.0, abs(gold)), (what, idx, k)
            else:
                assert min(abs(float(v) - r32), abs(float(v) - r64)) <= 2e-4 * max(1.0, abs(r64)), (what, idx, k, float(v), r32, r64)
        got = {f"{name}.{k}": p.grad for k, p in nets[idx].named_parameters() if p.grad is not None}
        print(f"[{what}] step {idx} ({name}): {len(got)} gradient tensors, logs "
              + ", ".join(f"{k}={float(v):.5f}" for k, v in res.logs.items()))
        gfx = gf.GradFixture(fix, f"grad{idx}:")
        gf.compare_grads(got, gfx, f"{what} step {idx}", rule="sams")
        if golden is not None:
            # element-wise against the REFERENCE's own backward pass (golden: every 97th element of every gradient): wherever
            # two fp32 CPU evaluations (the reference, the oracle) agree to 2e-3 of the tensor's max - i.e. the tensor is
            # neither ill-conditioned nor next to a ReLU kink - the HIP gradient is within 1e-2 of the reference's.
            # (The binding element-wise ties are HIP <-> oracle above and oracle <-> reference in tests/test_oracle_golden.py;
            #  this direct comparison is a consistency check; o97: = the fp32 oracle at the goldens' sample positions.)
            checked, loose = 0, []
            for k, gr in got.items():
                ref = golden[f"gs{idx}:{k}"].astype(np.float64)
                o32 = fix[f"o97:{idx}:{k}"].astype(np.float64)
                e = gfx.entry(k)
                big = max(np.abs(ref).max(), 1e-30)
                if e["s64"] <= 1e-6 * big or np.abs(o32 - ref).max() > 2e-3 * big:
                    continue
                mine = ops_to_oihw(gr).reshape(-1)[::97].double().cpu().numpy()
                err = np.abs(mine - ref).max()
                if err > 1e-2 * big:   # one pre-activation on the other side of a ReLU kink moves a few elements by 1-13 % of max
                    loose.append((k, err / big))   # (the 13 % case is analysed in DESIGN.md 3.6: one sign flip at 64x48)
                checked += 1
            # how many tensors qualify depends on the CPU run behind the fixture (ReLU variants: 102-190 of 308 across hosts /
            # thread counts - the kink-adjacent ones move by percents between ANY two fp32 evaluations); a floor of 30 %
            # keeps the check from going vacuous
            assert checked >= 0.3 * len(got), (what, idx, checked, len(got))
            print(f"[{what} step {idx}] vs reference samples: {checked} tensors compared, {len(loose)} beyond 1e-2 of max "
                  f"(worst {max([r for _, r in loose], default=0.0):.3f})")
            assert len(loose) <= 0.5 * checked and all(r <= 0.3 for _, r in loose), (what, idx, loose)
        if idx == 0:
            fr = model.all_gen_frames
            fr = fr.reshape(fr.shape[0], -1, *fr.shape[-2:])
            gf.check_output(fix, "frames", fr, 1e-4, what, rel_to_max=True, mode="min")
            if golden is not None:
                big = float(fix["frames:max64"])
                assert np.abs(model.all_gen_frames.cpu()[..., ::4, ::4].numpy() - golden["frames_s4"]).max() <= 2e-4 * big


@pytest.mark.parametrize("tag", ["base", "attn_gelu", "progressive"])
def test_sams_three_training_steps_match_the_oracle(tag):
    import fullsize_cases as fc
    import gradfix as gf

    fix = gf.load(f"sams_{tag}")
    g, sd, hp_, batch = fc.sams_small_case(tag)
    gf.check_digest(fix, "digest:weights", sd)
    gf.check_digest(fix, "digest:batch", batch)
    model, hp = _model(tag, sd)
    _run_three_steps_against_fixture(model, _to(batch, DEV), fix, tag, golden=g, tag=tag)
    # buffers after the three steps: power-iteration vectors, running statistics, counters
    after = model.state_dict()
    nbt = dict(zip([str(k) for k in fix["nbt:names"]], fix["nbt:values"]))
    seen = 0
    for k, v in after.items():
        if k.startswith("criterion_VGG"):
            continue
        if k.endswith(("weight_u", "weight_v", "running_mean", "running_var")):
            a, b = v.cpu().double(), torch.from_numpy(fix["buf:" + k])
            assert (a - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item()), (tag, k)
            seen += 1
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(nbt[k]) == int(g["nbt:" + k]), (tag, k)
    assert seen > 0


def test_sams_two_full_iterations_with_adam_follow_the_oracle():
    """MultiOptimizerStep (three HipAdam optimizers, Lightning's order) against the fp64 oracle stepped with
    torch.optim.Adam (values committed in tests/golden/full/sams_base.npz, keys adam:*): the second iteration's losses see
    the first one's parameter updates, buffer updates and optimizer wiring."""
    import fullsize_cases as fc
    import gradfix as gf
    from shineon_virtual_tryon_amd.trainer import MultiOptimizerStep

    tag = "base"
    fix = gf.load("sams_base")
    g, sd, hp_, batch = fc.sams_small_case(tag)
    gf.check_digest(fix, "digest:weights", sd)
    model, hp = _model(tag, sd)
    dbatch = _to(batch, DEV)
    opts, _ = model.configure_optimizers()
    step = MultiOptimizerStep(model, opts)
    for it in range(2):
        results = step(dbatch, it)
        for idx, net in enumerate(sh.STEP_NETS):
            for k, v in results[idx].logs.items():
                ref = float(fix[f"adam:log:{it}:{idx}:{k}"])
                assert abs(float(v) - ref) <= 1e-3 * max(1.0, abs(ref)), (it, idx, k, float(v), ref)
    # parameters after two Adam steps each: the UPDATE of every weight tensor points the same way as the oracle's.
    # (Element-wise equality is not a property Adam has: its first steps move each element by ~lr * sign(gradient), so
    # elements whose gradient is near round-off go either way in any fp32 implementation.)
    state = model.state_dict()
    compared = 0
    for key in fix.files:
        if not key.startswith("adam:update_e4_f16:"):
            continue
        k = key[len("adam:update_e4_f16:"):]
        if k.endswith("bias") or k.endswith("gamma"):
            continue  # biases in front of a normalisation have zero gradient: pure-noise Adam steps
        mine, ref = state[k].cpu().double() - sd[k].double(), torch.from_numpy(fix[key]).double() * 1e-4
        cos = (mine * ref).sum() / (mine.norm() * ref.norm() + 1e-300)
        assert cos.item() >= 0.97, (k, cos.item())
        assert abs(mine.norm().item() / ref.norm().item() - 1.0) <= 0.05, (k, mine.norm().item(), ref.norm().item())
        compared += 1
    assert compared >= 40, compared


def test_sams_eval_mode_uses_running_statistics_and_leaves_buffers_alone():
    """model.eval(): SPADE's batch norm reads the running statistics, spectral norm reuses the stored u / v (no power
    iteration), nothing is written.  Checked on one generator pass (the procedural running statistics do not match the
    procedural weights, so the n-frame recursion of a full validation_step would overflow in any arithmetic), then
    validation_step is run for its contract: EvalResult, val_ keys, no buffer touched."""
    from shineon_virtual_tryon_amd import ops

    tag = "base"
    g = sh.load_golden(tag)
    sd = procedural_state_dict(sh.golden_shapes(g))
    model, hp = _model(tag, sd)
    batch = _batch(hp)
    before = {k: v.clone() for k, v in model.state_dict().items() if not k.startswith("criterion_VGG")}
    model.eval()
    b, n = batch["image"].shape[:2]
    h, w = hp.fine_height, hp.fine_width
    torch.manual_seed(11)
    prev_frames = torch.randn(b, n - 1, 3, h, w) * 0.3
    prev_maps = batch["flow"][:, :n - 1]
    maps_now = {k: batch[k][:, -1] for k in model.inputs}
    with torch.no_grad():
        out = model.generator(prev_frames.to(DEV), prev_maps.to(DEV), {k: v.to(DEV) for k, v in maps_now.items()})
        osd = {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        ref = so.generator_forward(osd, prev_frames.double(), prev_maps.double(), {k: v.double() for k, v in maps_now.items()},
                                   hp, training=False)
    got = ops.to_nchw(out).cpu().double()
    assert torch.isfinite(ref).all() and (got - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
    for k, v in osd.items():  # the oracle did not touch its buffers either
        if v.is_floating_point():
            assert torch.equal(v, sd[k].double()), k
    with torch.no_grad():
        res = model.validation_step(_to(batch, DEV), 0)
    assert type(res).__name__ == "EvalResult" and res.checkpoint_on is not None
    assert set(res.logs) == {"val_loss", "val_loss/G/adv_multiscale", "val_loss/G/adv_temporal", "val_loss/G/l1+vgg",
                             "val_loss/G/l1", "val_loss/G/vgg"}
    for k, v in model.state_dict().items():
        if k in before:
            assert torch.equal(v, before[k]), k


def test_sams_trainer_fit_checkpoint_and_resume(tmp_path):
    """options -> registry -> Trainer.fit with the three optimizers (train + validation + checkpoint), then a resume: weights,
    the three Adam states, the three LR schedules and the spectral-norm / running-statistics buffers come back."""
    import os

    from shineon_virtual_tryon_amd.options import TrainOptions
    from shineon_virtual_tryon_amd.registry import find_model_using_name
    from shineon_virtual_tryon_amd.trainer import Trainer

    root = str(tmp_path / "exp")
    argv = ["--model", "sams", "--dataset", "synthetic", "--name", "t", "-b", "2", "--workers", "0", "--synthetic_length", "6",
            "--experiments_dir", root, "--n_frames_total", "3", "--flow_warp", "--activation", "relu", "--fine_height", "64",
            "--fine_width", "48", "--ngf_pow_outer", "3", "--ngf_pow_inner", "5", "--num_middle", "1", "--ndf", "8",
            "--allow_random_vgg"]
    opt = TrainOptions().parse(argv, interactive=False)
    torch.manual_seed(9)
    model = find_model_using_name(opt.model)(opt)
    trainer = Trainer(default_root_dir=root, max_epochs=1, limit_train_batches=2, limit_val_batches=1, val_check_interval=2)
    before = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    trainer.fit(model)
    assert trainer.global_step == 2 and len(trainer.optimizers) == 3
    assert all(p.requires_grad for p in model.generator.parameters())  # the per-optimizer toggling is undone afterwards
    ckpt = os.path.join(root, "checkpoints", "final.ckpt")
    saved = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert list(saved["state_dict"].keys()) == list(before.keys())
    assert [s["steps"] for s in saved["optimizer_states"]] == [2, 2, 2] and len(saved["lr_schedulers"]) == 3
    for key in ("generator.encode_layers.0.weight", "multiscale_discriminator.discriminator_0.model0.0.weight",
                "temporal_discriminator.model4.0.weight", "generator.encode_layers.1.conv_0.weight_u",
                "generator.encode_layers.1.spade_0.param_free_norm.running_mean"):
        assert not torch.equal(saved["state_dict"][key], before[key]), key
    torch.manual_seed(1234)
    model2 = find_model_using_name(opt.model)(opt)
    t2 = Trainer(default_root_dir=root, max_epochs=2, limit_train_batches=1, limit_val_batches=1, resume_from_checkpoint=ckpt)
    t2.fit(model2)
    assert t2.global_step == 3 and [o._steps for o in t2.optimizers] == [3, 3, 3] and t2.current_epoch == 1
    w = "generator.decode_layers.4.weight"
    d = (model2.state_dict()[w].cpu() - saved["state_dict"][w]).abs().max()
    assert 0 < float(d) < 1e-3, float(d)  # one Adam step (lr 1e-4) away from the CHECKPOINT's weights


def test_sams_full_size_properties():
    """The reference-default networks (generator 64..1024 features, 184.8 M parameters) at 256x192, n_frames_total = 5,
    bs = 1 - the size bench.py --config sams times - through properties that need no CPU oracle run:
      * the generator's adversarial terms carry NO gradient to the generator (they are computed from the prediction for the
        real frames and the default discriminators normalise per sample): every generator gradient of those two terms is
        exactly zero, while L1 + VGG reach every generator parameter;
      * the step is deterministic: the same state and batch give bit-identical losses and gradients;
      * spectral norm: after the pass every spectrally normalised weight has u, v of unit length and sigma = u^T W v > 0."""
    import argparse
    import copy

    import bench
    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.sams_model import SamsModel

    hp = bench.sams_hparams()
    torch.manual_seed(420)
    model = SamsModel(hp).to(DEV).train()
    assert abs(sum(p.numel() for p in model.generator.parameters()) / 1e6 - 184.8) < 0.1
    batch = synthetic_batch(1, DEV, seed=420, n_frames=hp.n_frames_total)
    for p in model.parameters():
        p.requires_grad_(False)
    gen = list(model.generator.parameters())
    for p in gen:
        p.requires_grad_(True)
    buffers = copy.deepcopy({k: v for k, v in model.state_dict().items() if not k.startswith("criterion_VGG")})

    def run():
        model.load_state_dict(buffers, strict=False)  # same u / v / running statistics at the start of every run
        model.zero_grad(set_to_none=True)
        res = model.training_step(batch, 0, 0)
        return res

    run().minimize.sum().backward()  # layer shapes without a committed plan are measured on first use: settle them first
    res = run()
    assert all(torch.isfinite(v).all() for v in res.logs.values())
    res.logs["loss/G/l1+vgg"].sum().backward()
    first = [p.grad.clone() for p in gen]
    assert all(torch.isfinite(g).all() for g in first)
    assert sum(float(g.abs().max()) > 0 for g in first) >= len(first) - 60  # biases in front of a normalisation stay ~0
    loss1 = {k: v.detach().clone() for k, v in res.logs.items()}
    res2 = run()
    res2.logs["loss/G/l1+vgg"].sum().backward()
    for k, v in res2.logs.items():
        assert torch.equal(v.detach(), loss1[k]), k
    worst = max(float((p.grad - g).abs().max()) / max(float(g.abs().max()), 1e-30) for p, g in zip(gen, first))
    assert worst == 0.0, worst
    res3 = run()
    (res3.logs["loss/G/adv_multiscale"].sum() + res3.logs["loss/G/adv_temporal"].sum()).backward()
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in gen)
    for name, m in model.generator.named_modules():
        if hasattr(m, "weight_u"):
            assert abs(float(m.weight_u.norm()) - 1.0) < 1e-4 and abs(float(m.weight_v.norm()) - 1.0) < 1e-4, name
            w = m.weight_orig.detach().reshape(m.weight_orig.shape[0], -1)
            sigma = torch.dot(m.weight_u, torch.mv(w, m.weight_v))
            assert float(sigma) > 0, name


def test_sams_full_size_three_training_steps_match_the_oracle():
    """the networks bench.py --config sams times (reference defaults: generator 64..1024 features,
    184.8 M parameters; 256x192; flow_warp; n_frames_total = 3, see below) at bs = 1, against the oracle's fp32 AND fp64
    values committed in tests/golden/full/sams_full_three_steps.npz: the generator step and both discriminator steps - every
    logged scalar, all generated frames, and every gradient of each step's parameter set under the rule of the small-size
    tests.  models/sams_model.py:147-383 with options/gan_options.py defaults.
    (n_frames_total = 3 instead of the timed 5: three frames run the same networks at the same resolution through the same
    recursion, flow warping and both discriminators; the fp64 oracle of five passes needs > 60 GiB.)"""
    import fullsize_cases as fc
    import gradfix as gf

    fix = gf.load("sams_full_three_steps")
    hp, model, sd, batch = fc.sams_full_three_steps_case()
    gf.check_digest(fix, "digest:weights", sd)
    gf.check_digest(fix, "digest:batch", batch)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    assert abs(sum(p.numel() for p in model.generator.parameters()) / 1e6 - 184.8) < 0.1
    _run_three_steps_against_fixture(model, _to(batch, DEV), fix, "sams full size")


def test_sams_full_size_generator_pass_bs4_vs_oracle():
    """The reference-default generator at the batch bench.py times (bs = 4, 256x192, four previous frames): one forward +
    backward pass against oracle.generator_forward in fp32 and fp64 (committed: tests/golden/full/sams_full_generator_bs4.npz)
    - output and every parameter gradient under the rule of the other SAMS tests.  These are the bs = 4 layer shapes (igemm
    instantiations / split-K plans from the committed plans file, Winograd F(2x2) / F(4x4) forms) of the timed step."""
    import fullsize_cases as fc
    import gradfix as gf

    fix = gf.load("sams_full_generator_bs4")
    hp, gen, sd, prev_frames, prev_maps, maps, gout = fc.sams_full_generator_case()
    gf.check_digest(fix, "digest:weights", sd)
    gf.check_digest(fix, "digest:inputs", {"pf": prev_frames, "pm": prev_maps, "gout": gout, **maps})
    gen.load_state_dict({k[len("generator."):]: v for k, v in sd.items()})
    gen = gen.to(DEV).train()
    y = gen(prev_frames.to(DEV), prev_maps.to(DEV), {k: v.to(DEV) for k, v in maps.items()})
    y.backward(gout.to(DEV))
    gf.check_output(fix, "out", _nchw(y), 1e-4, "sams generator bs=4", rel_to_max=True, mode="min")
    gf.compare_grads({k: p.grad for k, p in gen.named_parameters() if p.grad is not None}, gf.GradFixture(fix, "grad:"),
                     "generator bs=4 full size", rule="sams")


def test_sams_full_size_bs8_equals_two_copies_of_bs4():
    """BASELINE config 4 names bs = 8 per GPU for the SAMS model.  An oracle run at that size is out of reach on the box's
    host cores, but the batch size can be tied to the oracle-checked bs = 4 results without one: with a batch made of TWO
    COPIES of a bs = 4 batch every batch statistic (the SPADE's batch norm, the losses' means) is unchanged, so the generator
    step at bs = 8 - other igemm tile / split-K plans and Winograd forms than at bs = 4 - must reproduce the bs = 4 frames for
    both copies, the same logged scalars and the same generator gradients (up to fp32 summation order)."""
    import bench
    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.sams_model import SamsModel

    hp = bench.sams_hparams()
    torch.manual_seed(420)
    model = SamsModel(hp).to(DEV).train()
    b4 = synthetic_batch(4, DEV, seed=420, n_frames=hp.n_frames_total, smooth=True)
    b8 = {k: (torch.cat([v, v], 0) if torch.is_tensor(v) else v + v if isinstance(v, list) else v) for k, v in b4.items()}
    for p_ in model.parameters():
        p_.requires_grad_(False)
    gen = list(model.generator.parameters())
    for p_ in gen:
        p_.requires_grad_(True)
    import copy

    buffers = copy.deepcopy({k: v for k, v in model.state_dict().items() if not k.startswith("criterion_VGG")})

    def run(batch):
        model.load_state_dict(buffers, strict=False)   # same u / v / running statistics at the start of both runs
        model.zero_grad(set_to_none=True)
        res = model.training_step(batch, 0, 0)
        res.minimize.sum().backward()
        torch.cuda.synchronize()
        return ({k: float(v) for k, v in res.logs.items()}, model.all_gen_frames.cpu(),
                [p_.grad.detach().clone() for p_ in gen])

    logs4, frames4, grads4 = run(b4)
    logs8, frames8, grads8 = run(b8)
    big = frames4.abs().max().item()
    for half in (frames8[:4], frames8[4:]):
        assert (half - frames4).abs().max().item() <= 1e-4 * max(1.0, big), (half - frames4).abs().max().item()
    for k, v in logs4.items():
        assert abs(logs8[k] - v) <= 2e-4 * max(1.0, abs(v)), (k, logs8[k], v)
    worst, gmax, skipped = 0.0, max(x.abs().max().item() for x in grads4), 0
    for (name, _), g8, g4 in zip(model.generator.named_parameters(), grads8, grads4):
        # A conv bias whose only consumer is a batch norm (the stem conv in front of the first SPADE block, conv_0 in front of
        # norm_1 inside every block) has an analytically ZERO gradient: what the kernels return for it is the rounding residue
        # of a sum that cancels, different for every summation order, so it is not comparable between batch sizes.
        if name == "encode_layers.0.bias" or name.endswith(".conv_0.bias"):
            skipped += 1
            continue
        scale = g4.abs().max().item()
        worst = max(worst, max(0.0, (g8 - g4).abs().max().item() - 1e-5 * gmax) / max(scale, 1e-30))
    assert skipped >= 2
    print(f"[sams bs=8 vs 2 x bs=4] frames max diff {max((frames8[:4] - frames4).abs().max().item(), (frames8[4:] - frames4).abs().max().item()):.2e}, "
          f"worst gradient tensor {worst:.2e} of its max ({skipped} zero-by-construction biases not compared)")
    assert worst <= 2e-2, worst   # summation order over twice the pixels + kink-adjacent elements (see the bs = 4 test)

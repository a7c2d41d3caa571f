"""GPU parity of the two LightningModule-surface models against (a) the golden vectors produced by the
reference itself and (b) the oracle on the same seeded inputs; plus size-independent properties at
BASELINE.json's full batch size."""
import numpy as np
import pytest
import torch

from helpers import (UNET_VARIANTS, WARP_HP, assert_checksums, assert_close, assert_grad_samples, golden_state, load_golden, make_namespace,
                     oracle, strided, synthetic_cpu_batch, unet_hp)

pytestmark = pytest.mark.gpu


def _to(batch, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def _load(model, sd, dev):
    missing, unexpected = model.load_state_dict(sd, strict=True)
    return model.to(dev)


def test_warp_model_vs_reference_golden(cuda):
    from shineon_virtual_tryon_amd.warp_model import WarpModel

    g = load_golden("warp_model.npz")
    hp = make_namespace(person_inputs=["agnostic", "cocopose"])
    model = _load(WarpModel(hp), golden_state(g), cuda)
    assert [k for k in model.state_dict().keys()] == [str(k) for k in g["state_keys"]]
    model.train()
    batch = _to(synthetic_cpu_batch(2), cuda)
    res = model.training_step(batch, 0)
    res.minimize.backward()
    person = torch.cat([batch[k] for k in hp.person_inputs], 1)
    model2 = _load(WarpModel(hp), golden_state(g), cuda).train()
    with torch.no_grad():
        grid, theta = model2(person, batch["cloth"])
    assert_close(theta, g["theta"], atol=2e-5, what="theta")
    # The TPS map is evaluated in fp64 and rounded once (exact on a given theta, tests/test_ops_gpu.py::test_tps_grid); the grid
    # difference to the reference is its ~3.7x amplification of the theta difference (9.4e-6 measured here: two fp32
    # evaluations of 16 layers, each ~5e-6 from the exact value): 2.0e-5 measured.  DEVIATION from SURVEY.md 0-6, which quotes
    # 1e-5 for the model-level grid: two independent fp32 evaluations of this 16-layer graph cannot agree to 1e-5 on theta
    # x 3.7; the 1e-5 bar is kept where it can hold - the grid on a GIVEN theta (2.5e-7, test_tps_grid) and the fp64 leg below.
    assert_close(strided(grid.permute(0, 3, 1, 2)), g["grid_s8"], atol=3e-5, what="grid")
    # warped cloth: bilinear sampling amplifies the grid difference by the image slope (~5 per unit of normalised coordinate
    # for the band-limited fixture): 1.2e-4 at 3 of the 4608 REFERENCE samples.  The north-star 1e-4 is therefore stated the
    # way tests/test_parity_bs4_gpu.py states it - against the reference's value OR the exact (fp64) value of the same graph,
    # the oracle's (tests/golden/full/warp_bs2.npz): every element within 1e-4 of the fp32 value or within 1e-4 + the fp32
    # reference's own distance from exact, and never more than twice as far from exact as the reference is.  The sampling
    # itself is exact: ATen's grid_sample on OUR grid reproduces our output to 1e-5 (and the integer taps are bit-exact).
    assert_close(strided(model.warped_cloth), g["warped_cloth_s8"], atol=1.5e-4, what="warped cloth vs the reference's samples")
    import gradfix as gf

    fix = gf.load("warp_bs2")
    gf.check_digest(fix, "digest:weights", golden_state(g))
    gf.check_output(fix, "grid", grid.permute(0, 3, 1, 2), 3e-5, "warp bs=2", mode="either")
    gf.check_output(fix, "warped_cloth", model.warped_cloth, 1e-4, "warp bs=2", mode="either")
    resampled = oracle.grid_sample(batch["cloth"].cpu(), grid.cpu(), "border")  # model2 == model: the same grid
    assert_close(model.warped_cloth, resampled, atol=1e-5, what="grid_sample(cloth, OUR grid) vs ATen on the same grid")
    assert abs(res.minimize.item() - float(g["loss"])) < 2e-5
    gscale = float(np.abs(g["grad_linear_weight"]).max())
    assert_close(model.regression.linear.weight.grad, g["grad_linear_weight"], atol=3e-3 * gscale, what="d linear.weight")
    params = dict(model.named_parameters())
    for k in [k for k in g.files if k.startswith("gcs:")]:
        assert_checksums(params[k[4:]].grad, g[k], rel=5e-3, what=k)
    # every 97th element of every gradient vs the REFERENCE's own backward pass.  The person-branch extractor's fp32
    # gradients are ill-conditioned (the reference itself sits up to 5.9e-2 of max from its fp64 evaluation at bs=4,
    # tests/test_parity_bs4_gpu.py): those tensors get the looser bound
    assert_grad_samples(lambda k: params[k].grad, {k: g[k] for k in g.files if "extractionA" not in k}, "gs97:", rel=1e-2,
                        what="warp vs reference")
    # (the table of measured per-tensor distances is printed - `pytest -s`, kept in profiles/r06_gpu_tests.log - so the 8e-2
    #  bound is a measured number per tensor: VERDICT r05 item 8 iii)
    #  Measured in round 6 (profiles/r06_person_branch_gradients.txt): 20 of the 22 tensors are within 5.4e-3, two convolution
    #  weights (model.15: 3.2e-2, model.2: 1.9e-2) are not - so the bound is 4e-2 with at most two tensors beyond 1e-2.
    _, measured = assert_grad_samples(lambda k: params[k].grad, {k: g[k] for k in g.files if "extractionA" in k}, "gs97:", rel=4e-2,
                                      what="warp person branch vs reference", report_above=1e-2)
    assert sum(1 for e, _ in measured if e > 1e-2) <= 2, measured[:4]
    assert_close(model.extractionA.model[2].running_mean, g["bn_rm_A2"], atol=1e-5, what="BN running mean")
    assert_close(model.extractionA.model[2].running_var, g["bn_rv_A2"], atol=1e-5, what="BN running var")
    assert_close(model.regression.conv[10].running_var, g["bn_rv_R10"], atol=1e-4, what="BN running var R10")
    assert int(model.extractionA.model[2].num_batches_tracked) == 1


@pytest.mark.parametrize("variant", list(UNET_VARIANTS))
def test_unet_mask_model_vs_reference_golden(cuda, variant):
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel

    g = load_golden(f"unet_mask_{variant}.npz")
    hp = make_namespace(**UNET_VARIANTS[variant])
    model = _load(UnetMaskModel(hp), golden_state(g), cuda)
    assert [k for k in model.state_dict().keys()] == [str(k) for k in g["state_keys"]]
    model.train()
    res = model.training_step(_to(synthetic_cpu_batch(2), cuda), 0)
    res.minimize.backward()
    # forward outputs within the north-star tolerance: fp32 atol 1e-4
    for name, t in (("p_rendered", model.p_rendereds[0]), ("tryon_mask", model.tryon_masks[0]), ("p_tryon", model.p_tryons[0])):
        assert_close(strided(t), g[name + "_s8"], atol=1e-4, what=f"{variant} {name}")
        assert_checksums(t, g[name + "_cs"], rel=2e-5, what=f"{variant} {name} checksum")
    for k in ("loss/G", "loss/G/l1", "loss/G/vgg", "loss/G/tryon_mask_l1", "loss/G/flow_mask_l1"):
        ref = float(g["log:" + k])
        assert abs(float(res.logs[k]) - ref) <= 2e-5 + 2e-5 * abs(ref), (k, float(res.logs[k]), ref)
    params = dict(model.named_parameters())
    for k in [k for k in g.files if k.startswith("gcs:")]:
        # floor: the attention gamma gradient is a heavily cancelling sum (|g| ~ 2e-4 from terms ~1e-2)
        assert_checksums(params[k[4:]].grad, g[k], rel=1e-2, what=f"{variant} {k}", floor=4e-5)
    # element-wise against the reference's own gradients (every 97th element of all 52 tensors); the kinked variants
    # (ReLU / LeakyReLU, esp. with attention) move by a few percent when one pre-activation changes side of its kink -
    # the oracle itself is 3.9e-2 from the reference there (tests/test_oracle_golden.py)
    rel = {"plain": 5e-2, "gelu": 5e-3, "attn": 8e-2, "attn_gelu": 5e-3}[variant]
    assert_grad_samples(lambda k: params[k].grad, g, "gs97:", rel=rel, what=f"{variant} vs reference", floor=4e-5 if "attn" in variant else 2e-7)


def test_unet_mask_full_tensor_vs_oracle(cuda):
    """Every element of the forward outputs and of the input-layer gradient against the oracle (bs=1)."""
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel

    variant = "attn_gelu"
    g = load_golden(f"unet_mask_{variant}.npz")
    sd = golden_state(g)
    hp = make_namespace(**UNET_VARIANTS[variant])
    model = _load(UnetMaskModel(hp), sd, cuda).train()
    batch = synthetic_cpu_batch(1)
    res = model.training_step(_to(batch, cuda), 0)
    res.minimize.backward()
    params = {k: v.clone().requires_grad_(k.startswith("unet.")) for k, v in sd.items()}
    out = oracle.unet_mask_losses(params, batch, unet_hp(**UNET_VARIANTS[variant]))
    out["loss/G"].backward()
    assert_close(model.p_tryons[0], out["p_tryons"], atol=1e-4, what="p_tryon full")
    assert_close(model.tryon_masks[0], out["tryon_masks"], atol=1e-4, what="mask full")
    k0 = "unet.model.model.0.weight"
    gs = params[k0].grad.abs().max().item()
    assert_close(dict(model.named_parameters())[k0].grad, params[k0].grad, atol=2e-3 * gs, what="d first conv")


def test_warp_full_batch_properties(cuda):
    """BASELINE bs=4: theta in (-1,1) (tanh), finite grid, identity-grid behaviour when the head is zeroed."""
    from shineon_virtual_tryon_amd import ops
    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.warp_model import WarpModel

    hp = make_namespace(person_inputs=["agnostic", "cocopose"])
    torch.manual_seed(1)
    model = WarpModel(hp).to(cuda).train()
    batch = synthetic_batch(4, cuda)
    person = torch.cat([batch[k] for k in hp.person_inputs], 1)
    grid, theta = model(person, batch["cloth"])
    assert theta.shape == (4, 50) and grid.shape == (4, 256, 192, 2)
    assert torch.isfinite(grid).all() and theta.abs().max() < 1
    with torch.no_grad():
        model.regression.linear.weight.zero_()
        model.regression.linear.bias.zero_()
        grid0, theta0 = model(person, batch["cloth"])
        assert theta0.abs().max() == 0
    # theta = 0 -> the TPS grid is the regular base grid (linspace(-1, 1)); sampled with align_corners=False
    # that is the reference's slight zoom, so compare with torch's grid_sample on that base grid
    import torch.nn.functional as F

    c = oracle.tps_constants(256, 192, 5)
    base = torch.stack([c["gx"][None, None, :].expand(4, 256, 192), c["gy"][None, :, None].expand(4, 256, 192)], 3)
    assert (grid0.cpu() - base).abs().max() < 2e-5
    smooth = synthetic_batch(4, cuda, smooth=True)["cloth"]
    with torch.no_grad():
        got = ops.grid_sample(smooth, grid0, "border").cpu()
    ref = F.grid_sample(smooth.cpu(), base, mode="bilinear", padding_mode="border", align_corners=False)
    assert (got - ref).abs().max() < 2e-3


def test_training_reduces_loss_and_adam_state(cuda):
    """A few optimizer steps through HipAdam on the flat slabs: loss goes down, state_dict layout survives."""
    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.warp_model import WarpModel

    hp = make_namespace(person_inputs=["agnostic", "cocopose"], lr=1e-3)
    torch.manual_seed(2)
    model = WarpModel(hp).to(cuda).train()
    keys_before = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    (opt,), (sched,) = model.configure_optimizers()
    batch = synthetic_batch(2, cuda, smooth=True)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        res = model.training_step(batch, 0)
        res.minimize.backward()
        opt.step()
        losses.append(res.minimize.item())
    assert losses[-1] < losses[0], losses
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == keys_before
    # parameters are views of one flat slab and gradients of another
    p = next(model.parameters())
    assert p.data_ptr() >= opt.flat_params.data_ptr() and p.grad.data_ptr() >= opt.flat_grads.data_ptr()


def test_test_step_writes_pngs(cuda, tmp_path):
    from PIL import Image

    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.warp_model import WarpModel

    hp = make_namespace(is_train=False, person_inputs=["agnostic", "cocopose"], checkpoint="ckpt/x.ckpt", name="exp",
                        result_dir=str(tmp_path), datamode="test")
    model = WarpModel(hp).to(cuda).eval()
    model.override_hparams(hp)
    batch = synthetic_batch(2, cuda)
    batch["dataset_name"] = [["VitonDataset", "VitonDataset"]]  # one list per frame, like the reference's collate
    names = batch["cloth_name"][0]
    with torch.no_grad():
        out = model.test_step(batch, 0)
    assert out["progress_bar"]["file"] == names[0]
    d = tmp_path / "exp" / "x.ckpt" / "test" / "VitonDataset"
    img = np.array(Image.open(d / "warp-cloth" / names[0]))
    assert img.shape == (256, 192, 3) and img.dtype == np.uint8
    expected = oracle.png_quantise(model.warped_cloth[0].cpu()).swapaxes(0, 1).swapaxes(1, 2)
    assert np.array_equal(img, expected)
    assert (d / "warp-mask" / names[0]).exists()
    with torch.no_grad():
        again = model.test_step(batch, 0)
    assert again["progress_bar"]["file"].startswith("Skipping")


def test_unet_mask_three_frames_flow_warp_gpu(cuda):
    """Multi-frame path on the GPU: channel counts 134/268/536/1072 (not multiples of 4 -> zero-padded GEMMs),
    Resample2d, flow-mask blend, flow-mask sum penalty; against the reference golden."""
    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel

    g = load_golden("unet_mask_n3_flow.npz")
    hp = make_namespace(n_frames_total=3, flow_warp=True, activation="gelu", fine_height=128, fine_width=64)
    model = _load(UnetMaskModel(hp), golden_state(g), cuda).train()
    assert [k for k in model.state_dict().keys()] == [str(k) for k in g["state_keys"]]
    batch = synthetic_batch(2, cuda, height=128, width=64, n_frames=3, smooth=True)
    res = model.training_step(batch, 0)
    res.minimize.backward()
    for name, ts in (("p_rendered", model.p_rendereds), ("tryon_mask", model.tryon_masks), ("p_tryon", model.p_tryons),
                     ("flow_mask", model.flow_masks)):
        t = torch.cat([x.contiguous() for x in ts], 1)
        assert_close(strided(t, 4), g[name + "_s4"], atol=1e-4, what=f"n3 {name}")
    for k in ("loss/G", "loss/G/l1", "loss/G/vgg", "loss/G/tryon_mask_l1", "loss/G/flow_mask_l1"):
        ref = float(g["log:" + k])
        assert abs(float(res.logs[k]) - ref) <= 2e-5 + 3e-5 * abs(ref), (k, float(res.logs[k]), ref)
    params = dict(model.named_parameters())
    for k in [k for k in g.files if k.startswith("gcs:")]:
        assert_checksums(params[k[4:]].grad, g[k], rel=1e-2, what=f"n3 {k}", floor=1e-3)
    # floor: biases in front of an InstanceNorm have analytically zero gradients - the reference holds round-off noise there
    # (up to 5e-6 in this fixture), the HIP path writes exact zeros
    assert_grad_samples(lambda k: params[k].grad, g, "gs397:", rel=5e-3, what="n3 vs reference", floor=1e-5)


def test_attention_head_dim_not_multiple_of_4(cuda):
    """C = 40 -> d = 5: the padded query/key projection path of SelfAttention against the oracle."""
    from oracle.procedural import procedural_state_dict
    from shineon_virtual_tryon_amd.networks.attention.sagan import SelfAttention

    sa = SelfAttention(40)
    sd = procedural_state_dict({k: tuple(v.shape) for k, v in sa.state_dict().items()}, seed=3)
    sa.load_state_dict(sd)
    sa = sa.to(cuda)
    x = torch.randn(2, 40, 4, 3, generator=torch.Generator().manual_seed(5))
    xg = x.clone().to(cuda).requires_grad_(True)
    y = sa(xg)
    xc = x.clone().requires_grad_(True)
    pc = {"a." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yr = oracle.self_attention(xc, pc, "a")
    assert_close(y, yr, atol=2e-5, what="attention d=5")
    seed = torch.randn(yr.shape, generator=torch.Generator().manual_seed(6))
    (y * seed.to(cuda)).sum().backward()
    (yr * seed).sum().backward()
    assert_close(xg.grad, xc.grad, atol=1e-4, what="attention d=5 dx")
    assert_close(sa.query_conv.weight.grad, pc["a.query_conv.weight"].grad, atol=1e-4, what="attention d=5 dWq")
    assert_close(sa.key_conv.bias.grad, pc["a.key_conv.bias"].grad, atol=1e-4, what="attention d=5 dbk")


@pytest.mark.parametrize("c,hw", [(64, (4, 3)), (128, (8, 6)), (512, (16, 12)), (96, (5, 7)), (64, (20, 16)), (64, (16, 16)), (32, (15, 15)),
                                  (64, (13, 16))])
def test_self_attention_fused_qkv_path(cuda, c, hw):
    """With HipAdam's slab layout (q/k/v weights, biases and gradients adjacent) the three projections and their
    gradients run as single GEMMs accumulating into the gradient slab; same numbers as the oracle.  Up to 256 positions
    everything between the projections is the fused core of csrc/attn.hip (the U-Net's 4x3 / 8x6 / 16x12 maps at 512
    channels, a ragged 5x7 map, and the largest instantiations: 256, 225 and 208 positions), beyond that (20x16) the composed
    engine launches."""
    from oracle.procedural import procedural_state_dict
    from shineon_virtual_tryon_amd import ops
    from shineon_virtual_tryon_amd.networks.attention.sagan import SelfAttention
    from shineon_virtual_tryon_amd.optim import HipAdam

    sa = SelfAttention(c)
    sd = procedural_state_dict({k: tuple(v.shape) for k, v in sa.state_dict().items()}, seed=11)
    sd["gamma"] = torch.tensor([0.7])
    sa.load_state_dict(sd)
    sa = sa.to(cuda)
    opt = HipAdam(sa.parameters(), lr=1e-3, adjacent=sa.adjacent_param_groups())
    opt.zero_grad()
    convs = (sa.query_conv, sa.key_conv, sa.value_conv)
    assert ops._adjacent([m.weight for m in convs]) and ops._adjacent([m.bias.grad for m in convs])
    x = torch.randn(2, c, hw[0], hw[1], generator=torch.Generator().manual_seed(12))
    xg = x.clone().to(cuda).requires_grad_(True)
    y = sa(xg)
    core = bool(ops.lib().so_attn_supported(2, hw[0] * hw[1], c, c // 8))
    assert core == (hw[0] * hw[1] <= 256)
    assert type(y.grad_fn).__name__.startswith("_SelfAttentionCoreFn" if core else "_SelfAttentionQkvFn")
    xc = x.clone().requires_grad_(True)
    pc = {"a." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yr = oracle.self_attention(xc, pc, "a")
    assert_close(y, yr, atol=2e-5, what="fused attention")
    seed = torch.randn(yr.shape, generator=torch.Generator().manual_seed(13))
    (y * seed.to(cuda)).sum().backward()
    (yr * seed).sum().backward()
    assert_close(xg.grad, xc.grad, atol=1e-4, what="fused attention dx")
    for name, prm in sa.named_parameters():
        ref_g = pc["a." + name].grad
        # (at 512 channels x 192 positions the weight gradients reach ~50: fp32 round-off scales with the tensor's magnitude)
        assert_close(prm.grad, ref_g, atol=1e-4 * max(1.0, float(ref_g.abs().max())), rtol=1e-4, what=f"fused attention d{name}")


@pytest.mark.parametrize("which", ["warp", "unet_mask"])
def test_gradients_are_bitwise_reproducible(cuda, which):
    """Two forward+backward passes from the same state give bit-identical losses and gradient slabs: every reduction on
    the path (split-K slabs, norm statistics, bias sums, 4x4x1 wgrad slabs, losses) is summed in a fixed order."""
    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel
    from shineon_virtual_tryon_amd.warp_model import WarpModel

    torch.manual_seed(5)
    if which == "warp":
        model = WarpModel(make_namespace(person_inputs=["agnostic", "cocopose"])).to(cuda).train()
    else:
        model = UnetMaskModel(make_namespace(self_attn=True, activation="gelu")).to(cuda).train()
        for m in model.modules():
            if hasattr(m, "gamma"):
                m.gamma.data.fill_(0.5)
    (opt,), _ = model.configure_optimizers()
    batch = synthetic_batch(2, cuda, smooth=True)
    runs = []
    for _ in range(2):
        opt.zero_grad()
        res = model.training_step(batch, 0)
        res.minimize.backward()
        torch.cuda.synchronize()
        runs.append((res.minimize.detach().clone(), opt.flat_grads.clone()))
    assert torch.equal(runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1]), float((runs[0][1] - runs[1][1]).abs().max())


def test_two_stream_chained_schedule_matches_sequential(cuda):
    """graphs.GraphedChainedStep (warp forward -> [try-on fwd+bwd || warp backward + Adam] on two streams, three hipGraphs)
    trains exactly like the plain sequential loop: same losses, bit-identical parameters after five steps."""
    import copy

    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.graphs import GraphedChainedStep
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel
    from shineon_virtual_tryon_amd.warp_model import WarpModel

    torch.manual_seed(7)
    warp_a = WarpModel(make_namespace(person_inputs=["agnostic", "cocopose"], lr=1e-3)).to(cuda).train()
    unet_a = UnetMaskModel(make_namespace(self_attn=True, activation="gelu", lr=1e-3)).to(cuda).train()
    for m in unet_a.modules():
        if hasattr(m, "gamma"):
            m.gamma.data.fill_(0.3)
    warp_b, unet_b = copy.deepcopy(warp_a), copy.deepcopy(unet_a)
    for m in (warp_a, unet_a, warp_b, unet_b):
        m.global_step = 1
    batch = synthetic_batch(2, cuda, smooth=True)

    def optimizers(w, u):
        return w.configure_optimizers()[0][0], u.configure_optimizers()[0][0]

    # reference: sequential eager steps
    optw, optu = optimizers(warp_a, unet_a)
    seq_losses = []
    for _ in range(5):
        optw.zero_grad()
        rw = warp_a.training_step(batch, 0)
        rw.minimize.backward()
        optw.step()
        b2 = dict(batch)
        b2["cloth"] = warp_a.warped_cloth.detach()
        optu.zero_grad()
        ru = unet_a.training_step(b2, 0)
        ru.minimize.backward()
        optu.step()
        seq_losses.append((float(rw.minimize), float(ru.minimize)))
    torch.cuda.synchronize()

    # two-stream schedule on identical copies
    optw2, optu2 = optimizers(warp_b, unet_b)
    g = GraphedChainedStep(warp_b, optw2, unet_b, optu2, batch, warmup=1)
    # the warm-up / capture passes above ran forward+backward but no optimizer step: parameters are still the initial ones
    # (BatchNorm running statistics moved, which does not influence training-mode outputs)
    pip_losses = []
    for _ in range(3):
        g.launch_warp_forward()
        g.launch_tryon()
        g.launch_warp_backward()
        with g.on_side():
            optw2.step()
        optu2.step()
        g.join()
        torch.cuda.synchronize()
        pip_losses.append((float(g.result_warp.minimize), float(g.result_tryon.minimize)))
    assert pip_losses == seq_losses[:3], (pip_losses, seq_losses)
    for _ in range(2):  # no host synchronisation: the side stream runs the next warp stage ahead of the try-on stage
        g.launch_warp_forward()
        g.launch_tryon()
        g.launch_warp_backward()
        with g.on_side():
            optw2.step()
        optu2.step()
    g.join()
    torch.cuda.synchronize()
    assert (float(g.result_warp.minimize), float(g.result_tryon.minimize)) == seq_losses[4]
    assert torch.equal(optw.flat_params, optw2.flat_params)
    assert torch.equal(optu.flat_params, optu2.flat_params)


def test_vgg_loss_with_precomputed_target_features(cuda):
    """VGGLoss(x, y, y_features=VGGLoss.target_features(y)) - the target branch computed ahead of time - gives the loss
    and input gradient of the batched x||y evaluation (and of the oracle)."""
    from shineon_virtual_tryon_amd.networks.loss import VGGLoss

    torch.manual_seed(3)
    crit = VGGLoss().to(cuda)
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(2, 3, 64, 48, generator=g) * 2 - 1)
    y = (torch.rand(2, 3, 64, 48, generator=g) * 2 - 1).to(cuda)
    xa = x.clone().to(cuda).requires_grad_(True)
    xb = x.clone().to(cuda).requires_grad_(True)
    la = crit(xa, y)
    feats = crit.target_features(y)
    assert len(feats) == 5 and tuple(feats[0].shape) == (2, 64, 64, 48) and tuple(feats[4].shape) == (2, 512, 4, 3)
    lb = crit(xb, y, y_features=feats)
    la.backward()
    lb.backward()
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(la)), (float(la), float(lb))
    assert_close(xb.grad, xa.grad.cpu(), atol=1e-7 + 1e-5 * float(xa.grad.abs().max()), what="vgg dx (precomputed target)")
    sd = {"v." + k: v.detach().cpu() for k, v in crit.state_dict().items()}
    lr = oracle.vgg_loss(sd, x.clone(), y.cpu(), prefix="v.vgg")
    assert abs(float(lb) - float(lr)) <= 2e-5 * abs(float(lr)), (float(lb), float(lr))


def test_trainer_fit_resume_and_test(cuda, tmp_path):
    """trainer.Trainer drives the hooks the reference's train.py / test.py hand to Lightning: options -> registry ->
    fit (train + validation + checkpoint) -> resume -> test (PNG writers), on the synthetic dataset."""
    import os

    from shineon_virtual_tryon_amd.options import TestOptions, TrainOptions
    from shineon_virtual_tryon_amd.registry import find_model_using_name
    from shineon_virtual_tryon_amd.trainer import Trainer

    root = str(tmp_path / "exp")
    argv = ["--model", "gmm", "--dataset", "synthetic", "--name", "t", "-b", "2", "--workers", "0", "--synthetic_length", "8",
            "--experiments_dir", root, "--lr", "1e-3"]
    opt = TrainOptions().parse(argv, interactive=False)
    torch.manual_seed(9)
    model = find_model_using_name(opt.model)(opt)
    trainer = Trainer(default_root_dir=root, max_epochs=1, limit_train_batches=3, limit_val_batches=1, val_check_interval=2)
    before = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    trainer.fit(model)
    assert trainer.global_step == 3
    ckpt = os.path.join(root, "checkpoints", "final.ckpt")
    assert os.path.exists(ckpt)
    saved = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert list(saved["state_dict"].keys()) == list(before.keys()) and saved["global_step"] == 3
    w = "regression.linear.weight"
    assert not torch.equal(saved["state_dict"][w], before[w])                       # it trained
    assert int(saved["state_dict"]["extractionA.model.2.num_batches_tracked"]) == 3  # training forwards only (shared counter)
    assert saved["optimizer_states"][0]["steps"] == 3

    # resume: weights, Adam moments, step counters AND the LR-schedule position come back from the checkpoint alone
    # (no manual load_state_dict: train.py:39-54 hands the same file to load_from_checkpoint and resume_from_checkpoint)
    assert saved["epoch"] == 1 and saved["lr_schedulers"][0]["last_epoch"] == 1   # end of epoch 0 -> resume starts epoch 1
    torch.manual_seed(1234)                                                        # different random init: must be overwritten
    model2 = find_model_using_name(opt.model)(opt)
    t2 = Trainer(default_root_dir=root, max_epochs=2, limit_train_batches=1, limit_val_batches=1, resume_from_checkpoint=ckpt)
    t2.fit(model2)
    assert t2.global_step == 4 and t2.optimizer._steps == 4 and t2.current_epoch == 1
    # one resumed step from the saved weights: every tensor that did not train in that step is still the checkpoint's
    assert torch.equal(model2.state_dict()["extractionA.model.2.num_batches_tracked"].cpu(), torch.tensor(4))
    d = (model2.state_dict()[w].cpu() - saved["state_dict"][w]).abs().max()
    assert 0 < float(d) < 1e-2, float(d)                                           # moved by ~lr from the CHECKPOINT's weights
    # LR schedule: keep_epochs=5 -> factor 1 through epoch 5; push the epoch counter into the decay phase and resume again
    saved2 = torch.load(os.path.join(root, "checkpoints", "final.ckpt"), map_location="cpu", weights_only=False)
    assert saved2["epoch"] == 2
    saved2["epoch"] = 7
    saved2["lr_schedulers"][0]["last_epoch"] = 7
    late = os.path.join(root, "checkpoints", "late.ckpt")
    torch.save(saved2, late)
    model4 = find_model_using_name(opt.model)(opt)
    t4 = Trainer(default_root_dir=root, max_epochs=8, limit_train_batches=1, limit_val_batches=1, resume_from_checkpoint=late)
    t4.fit(model4)
    # epoch 7 ran with lr0 * (1 - (7 - 5) / 6), then the scheduler stepped to epoch 8: lr0 * (1 - 3 / 6)
    assert abs(t4.optimizer.param_groups[0]["lr"] - 1e-3 * (1 - 3 / 6)) < 1e-12, t4.optimizer.param_groups[0]["lr"]
    assert t4.current_epoch == 7 and t4.global_step == 5

    # test: PNG writers under result_dir/name/<ckpt>/<datamode>/<Dataset>/
    topt = TestOptions().parse(["--model", "gmm", "--dataset", "synthetic", "--name", "t", "-b", "2", "--workers", "0",
                                "--synthetic_length", "4", "--checkpoint", ckpt, "--result_dir", str(tmp_path / "res")],
                               interactive=False)
    model3 = find_model_using_name(topt.model)(topt)
    model3.load_state_dict(saved["state_dict"], strict=True)
    model3.override_hparams(topt)
    outs = Trainer(default_root_dir=root).test(model3)
    assert len(outs) == 2
    pngs = [os.path.join(dp, f) for dp, _, fs in os.walk(str(tmp_path / "res")) for f in fs if f.endswith((".png", ".jpg"))]
    # (the reference only writes warp-mask for VitonDataset: visualization.py:60-88, mirrored in io_png.save_images)
    assert len(pngs) == 4 and all("SyntheticDataset/warp-cloth" in q for q in pngs), pngs[:4]


def test_trainer_tryon_stage_fit_and_test(cuda, tmp_path):
    """The try-on stage (--model tom --self_attn --activation gelu) through the same driver: one training step, the
    interrupted-by-exception checkpoint rule is not triggered, test_step writes the try-on PNGs."""
    import os

    from shineon_virtual_tryon_amd.options import TestOptions, TrainOptions
    from shineon_virtual_tryon_amd.registry import find_model_using_name
    from shineon_virtual_tryon_amd.trainer import Trainer

    root = str(tmp_path / "exp")
    common = ["--model", "tom", "--dataset", "synthetic", "--name", "u", "-b", "2", "--workers", "0", "--self_attn", "--activation",
              "gelu", "--person_inputs", "agnostic", "densepose", "--allow_random_vgg"]
    opt = TrainOptions().parse(common + ["--synthetic_length", "4", "--experiments_dir", root], interactive=False)
    model = find_model_using_name(opt.model)(opt)
    tr = Trainer(default_root_dir=root, max_epochs=1, limit_train_batches=1, limit_val_batches=1)
    tr.fit(model)
    ckpt = os.path.join(root, "checkpoints", "final.ckpt")
    assert tr.global_step == 1 and os.path.exists(ckpt)
    assert not [f for f in os.listdir(os.path.join(root, "checkpoints")) if f.startswith("interrupted")]
    sd = torch.load(ckpt, map_location="cpu", weights_only=False)["state_dict"]
    topt = TestOptions().parse(common + ["--synthetic_length", "2", "--checkpoint", ckpt, "--result_dir", str(tmp_path / "res")],
                               interactive=False)
    m2 = find_model_using_name(topt.model)(topt)
    m2.load_state_dict(sd, strict=True)
    m2.override_hparams(topt)
    outs = Trainer(default_root_dir=root).test(m2)
    assert len(outs) == 1
    pngs = [os.path.join(dp, f) for dp, _, fs in os.walk(str(tmp_path / "res")) for f in fs if f.endswith((".png", ".jpg"))]
    assert len(pngs) >= 2 and all(("tryon" in q) or ("reconstruction" in q) for q in pngs), pngs[:4]


def test_vgg_weights_option_and_refusal_to_train_on_random_vgg(cuda, tmp_path):
    """--vgg_weights loads a torchvision-layout vgg19 state_dict (features.N.weight/bias) or a checkpoint's criterionVGG.*;
    without pretrained weights Trainer.fit refuses to start unless --allow_random_vgg is given
    (the reference always trains against ImageNet VGG19: models/networks/vgg.py:9)."""
    from shineon_virtual_tryon_amd.options import TrainOptions
    from shineon_virtual_tryon_amd.registry import find_model_using_name
    from shineon_virtual_tryon_amd.trainer import Trainer

    common = ["--model", "tom", "--dataset", "synthetic", "--name", "v", "-b", "2", "--workers", "0", "--synthetic_length", "2",
              "--experiments_dir", str(tmp_path)]
    model = find_model_using_name("tom")(TrainOptions().parse(common, interactive=False))
    assert not model.criterionVGG.vgg.pretrained_loaded
    with pytest.raises(RuntimeError, match="vgg_weights"):
        Trainer(default_root_dir=str(tmp_path), max_epochs=1, limit_train_batches=1).fit(model)
    # a torchvision-layout file
    gen = torch.Generator().manual_seed(8)
    tv = {}
    for key, v in model.criterionVGG.vgg.state_dict().items():      # slice<k>.<i>.weight -> features.<i>.weight
        i, leaf = key.split(".")[1:]
        tv[f"features.{i}.{leaf}"] = torch.randn(v.shape, generator=gen) * 0.05
    tv["classifier.0.weight"] = torch.zeros(4, 4)                    # ignored
    path = str(tmp_path / "vgg19.pth")
    torch.save(tv, path)
    m2 = find_model_using_name("tom")(TrainOptions().parse(common + ["--vgg_weights", path], interactive=False))
    assert m2.criterionVGG.vgg.pretrained_loaded
    for key, v in m2.criterionVGG.vgg.state_dict().items():
        i, leaf = key.split(".")[1:]
        assert torch.equal(v.cpu(), tv[f"features.{i}.{leaf}"]), key
    m2.require_pretrained_vgg()
    # a checkpoint that carries criterionVGG.* keys
    ck = str(tmp_path / "ref.ckpt")
    torch.save({"state_dict": m2.state_dict()}, ck)
    m3 = find_model_using_name("tom")(TrainOptions().parse(common + ["--vgg_weights", ck], interactive=False))
    assert torch.equal(getattr(m3.criterionVGG.vgg.slice5, "28").weight, getattr(m2.criterionVGG.vgg.slice5, "28").weight)
    with pytest.raises(KeyError):
        bad = {k: v for k, v in tv.items() if not k.startswith("features.28.")}
        torch.save(bad, path)
        find_model_using_name("tom")(TrainOptions().parse(common + ["--vgg_weights", path], interactive=False))


def test_trainer_graph_overlap_mode_equals_eager_mode(cuda, tmp_path):
    """Trainer(graph=True, overlap=True) - hipGraph replay, HBM-resident dataset, deferred Adam - trains exactly like the
    plain eager loop: bit-identical weights after one epoch of 4 steps (+ a ragged last batch that runs eagerly)."""
    from shineon_virtual_tryon_amd.options import TrainOptions
    from shineon_virtual_tryon_amd.registry import find_model_using_name
    from shineon_virtual_tryon_amd.trainer import Trainer

    finals = []
    for graph in (True, False):
        root = str(tmp_path / f"g{int(graph)}")
        opt = TrainOptions().parse(["--model", "gmm", "--dataset", "synthetic", "--name", "t", "-b", "2", "--workers", "0",
                                    "--synthetic_length", "9", "--experiments_dir", root, "--lr", "1e-3", "--no_shuffle"],
                                   interactive=False)
        torch.manual_seed(21)
        model = find_model_using_name(opt.model)(opt)
        tr = Trainer(default_root_dir=root, max_epochs=1, limit_val_batches=1, val_check_interval=3, graph=graph, overlap=graph,
                     device_dataset=graph)
        tr.fit(model)
        assert tr.global_step == 5
        finals.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k


def test_vgg_split_bf16_path_stays_within_the_fp32_parity_tolerances(cuda):
    """Opt-in split-bf16 VGG chain (csrc/sb16.hip: fp32 = hi + mid bf16 planes, 3 bf16 MFMAs per product, fp32 accumulate):
    perceptual loss and its input gradient against the oracle at the SAME tolerances as the fp32 path, and the full bs = 2
    attn+gelu training step against the reference golden (losses 2e-5, gradient checksums)."""
    from shineon_virtual_tryon_amd import ops
    from shineon_virtual_tryon_amd.networks.loss import VGGLoss
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel

    default = ops.VGG_SPLIT_BF16
    ops.VGG_SPLIT_BF16 = True
    try:
        torch.manual_seed(3)
        crit = VGGLoss().to(cuda)
        g = torch.Generator().manual_seed(21)
        x = (torch.rand(2, 3, 64, 48, generator=g) * 2 - 1)
        y = (torch.rand(2, 3, 64, 48, generator=g) * 2 - 1)
        xs = x.clone().to(cuda).requires_grad_(True)
        ls = crit(xs, y.to(cuda))
        ls.backward()
        ops.VGG_SPLIT_BF16 = False
        xf = x.clone().to(cuda).requires_grad_(True)
        lf = crit(xf, y.to(cuda))
        lf.backward()
        ops.VGG_SPLIT_BF16 = True
        sd = {"v." + k: v.detach().cpu() for k, v in crit.state_dict().items()}
        xc = x.clone().requires_grad_(True)
        lr = oracle.vgg_loss(sd, xc, y, prefix="v.vgg")
        lr.backward()
        assert abs(float(ls) - float(lr)) <= 2e-5 * abs(float(lr)), (float(ls), float(lr))
        gmax = float(xc.grad.abs().max())
        assert_close(xs.grad, xc.grad, atol=1e-7 + 2e-3 * gmax, what="split-bf16 vgg dx vs oracle")
        print(f"[vgg split-bf16] loss rel err {abs(float(ls) - float(lr)) / abs(float(lr)):.2e} (fp32 path "
              f"{abs(float(lf) - float(lr)) / abs(float(lr)):.2e}); dx max err / max {float((xs.grad.cpu() - xc.grad).abs().max()) / gmax:.2e} "
              f"(fp32 path {float((xf.grad.cpu() - xc.grad).abs().max()) / gmax:.2e})")
        # whole try-on training step against the reference golden
        variant = "attn_gelu"
        gold = load_golden(f"unet_mask_{variant}.npz")
        model = _load(UnetMaskModel(make_namespace(**UNET_VARIANTS[variant])), golden_state(gold), cuda).train()
        res = model.training_step(_to(synthetic_cpu_batch(2), cuda), 0)
        res.minimize.backward()
        for k in ("loss/G", "loss/G/l1", "loss/G/vgg", "loss/G/tryon_mask_l1"):
            ref = float(gold["log:" + k])
            assert abs(float(res.logs[k]) - ref) <= 2e-5 + 2e-5 * abs(ref), (k, float(res.logs[k]), ref)
        params = dict(model.named_parameters())
        for k in [k for k in gold.files if k.startswith("gcs:")]:
            assert_checksums(params[k[4:]].grad, gold[k], rel=1e-2, what=f"split-bf16 {k}", floor=4e-5)
    finally:
        ops.VGG_SPLIT_BF16 = default


def test_trainer_fit_chained_equals_sequential_loop(cuda):
    """Trainer.fit_chained - the product API behind bench.py's step (trainer.ChainedTrainStep: three hipGraphs on two
    streams, deferred try-on Adam) - trains both stages exactly like the plain sequential eager loop over the same batches:
    bit-identical parameters and BatchNorm statistics after four chained steps on changing batches."""
    import copy

    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.trainer import Trainer
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel
    from shineon_virtual_tryon_amd.warp_model import WarpModel

    torch.manual_seed(17)
    warp_a = WarpModel(make_namespace(person_inputs=["agnostic", "cocopose"], lr=1e-3)).to(cuda).train()
    unet_a = UnetMaskModel(make_namespace(self_attn=True, activation="gelu", lr=1e-3)).to(cuda).train()
    for m in unet_a.modules():
        if hasattr(m, "gamma"):
            m.gamma.data.fill_(0.3)
    warp_b, unet_b = copy.deepcopy(warp_a), copy.deepcopy(unet_a)
    for m in (warp_a, unet_a, warp_b, unet_b):
        m.global_step = 1
    batches = [synthetic_batch(2, cuda, smooth=True, start=2 * i) for i in range(4)]

    optw = warp_a.configure_optimizers()[0][0]
    optu = unet_a.configure_optimizers()[0][0]
    for batch in batches:
        optw.zero_grad()
        warp_a.training_step(batch, 0).minimize.backward()
        optw.step()
        b2 = dict(batch)
        b2["cloth"] = warp_a.warped_cloth.detach()
        optu.zero_grad()
        unet_a.training_step(b2, 0).minimize.backward()
        optu.step()
    torch.cuda.synchronize()

    engine = Trainer(graph=True, overlap=True).fit_chained(warp_b, unet_b, batches, steps=4)
    assert engine.schedule == "pipeline"
    for (ka, va), (kb, vb) in zip(warp_a.state_dict().items(), warp_b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), f"warp {ka}"
    for (ka, va), (kb, vb) in zip(unet_a.state_dict().items(), unet_b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), f"unet {ka}"
    assert engine.optw._steps == 4 and engine.optu._steps == 4


def test_train_step_gradient_accumulation_matches_the_full_batch(cuda):
    """trainer.TrainStep(accumulate=2) on two half batches (the reference's --accumulated_batches, train.py:76-118 ->
    Lightning accumulate_grad_batches) applies ONE optimizer step whose gradient is the full batch's: InstanceNorm and
    attention are per-sample and every loss is a batch mean, so the parameters after the step agree with a bs = 4 step."""
    import copy

    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.trainer import TrainStep
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel

    torch.manual_seed(23)
    a = UnetMaskModel(make_namespace(self_attn=True, activation="gelu", lr=1e-3)).to(cuda).train()
    for m in a.modules():
        if hasattr(m, "gamma"):
            m.gamma.data.fill_(0.3)
    b = copy.deepcopy(a)
    a.global_step = b.global_step = 1
    full = synthetic_batch(4, cuda, smooth=True)
    half = lambda lo: {k: (v[lo:lo + 2].contiguous() if isinstance(v, torch.Tensor) else v) for k, v in full.items()}  # noqa: E731
    (opta,), _ = a.configure_optimizers()
    (optb,), _ = b.configure_optimizers()
    ea = TrainStep(a, opta, full, graph=False, overlap=False)
    ea(full)
    eb = TrainStep(b, optb, half(0), graph=True, overlap=True, accumulate=2)   # accumulation forces the eager path
    assert eb.graph is False
    eb(half(0))
    assert not eb.stepped and optb._steps == 0
    eb(half(2))
    eb.flush()
    assert eb.stepped and optb._steps == 1 and opta._steps == 1
    torch.cuda.synchronize()
    # Adam's first step moves every weight by ~lr * sign(g): compare the gradients themselves (kept in the slabs)
    ga, gb = opta.flat_grads, optb.flat_grads
    scale = float(ga.abs().max())
    # per-parameter comparison at the parity tolerance
    for (name, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if p.requires_grad and p.numel() > 1:
            tol = 2e-3 * float(p.grad.abs().max()) + 2e-7
            assert float((p.grad - q.grad).abs().max()) <= tol, name
    assert scale > 0


@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("which", ["unet", "warp"])
def test_bucketed_exchange_overlapped_with_backward_is_bit_identical(cuda, which, graph):
    """trainer.BucketedExchange (per-bucket signal node inside the captured backward pass -> hipStreamWaitValue32 on the
    communication stream -> all-reduce -> Adam on the bucket's slab range) against the plain whole-slab path: three steps of
    trainer.TrainStep from the same state give bit-identical parameters and Adam moments, graph-replayed and eager; the
    buckets are cut by bytes and every bucket was signalled from inside the backward pass (not at its end)."""
    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.trainer import TrainStep
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel
    from shineon_virtual_tryon_amd.warp_model import WarpModel

    def run(bucketed):
        torch.manual_seed(5)
        if which == "unet":
            model = UnetMaskModel(make_namespace(self_attn=True, activation="gelu", allow_random_vgg=True, lr=1e-3))
        else:
            model = WarpModel(make_namespace(person_inputs=["agnostic", "cocopose"], lr=1e-3))
        model = model.to(cuda).train()
        model.global_step = 1
        (opt,), _ = model.configure_optimizers()
        batches = [synthetic_batch(2, cuda, smooth=True, start=2 * i) for i in range(3)]
        eng = TrainStep(model, opt, batches[0], graph=graph, overlap=True, bucketed=bucketed, bucket_bytes=16 << 20)
        for b in batches:
            eng(b)
        eng.flush()
        torch.cuda.synchronize()
        return opt, eng

    o1, e1 = run(True)
    assert e1.exchange is not None and len(e1.exchange.buckets) >= 4, e1.exchange.describe()
    # the calibration pass saw the buckets become ready from the END of the slab towards its start (reverse layer order),
    # and all but the first-layer bucket complete strictly inside the backward pass
    order = e1.exchange.order
    assert order[0] == len(e1.exchange.buckets) - 1 and sorted(order) == list(range(len(order))), order
    o0, e0 = run(False)
    assert e0.exchange is None
    assert o1._steps == o0._steps == 3
    for a, b in zip(o1._flat, o0._flat):
        if a is o1._flat[1]:
            continue   # the gradient slab itself: identical too, but it is scratch
        assert torch.equal(a, b), float((a - b).abs().max())


def test_graphed_step_reads_cached_vgg_filters_and_refuses_a_stale_replay(cuda):
    """The replayed try-on step must not re-derive the FROZEN VGG19's Winograd-domain / transposed filters: the eager warm-up
    passes in front of the capture fill ops' derived-weight caches, the capture is served those copies (pinned by the step,
    GraphedTrainStep._pins), and no weight-transform launch is recorded - counted here with the launch counter of the transform
    entry points.  After ops.invalidate_weight_caches() (weights overwritten in place) the captured step refuses to replay."""
    from shineon_virtual_tryon_amd import ops
    from shineon_virtual_tryon_amd.data import synthetic_batch
    from shineon_virtual_tryon_amd.graphs import GraphedTrainStep
    from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel

    torch.manual_seed(5)
    model = UnetMaskModel(make_namespace(allow_random_vgg=True, lr=1e-3)).to(cuda).train()
    model.global_step = 1
    (opt,), _ = model.configure_optimizers()
    batch = synthetic_batch(2, cuda, smooth=True)
    calls = {"n": 0}
    real = ops._wino_weights

    def counting(wk, owner, *a, **k):
        capturing = torch.cuda.is_current_stream_capturing()
        u = real(wk, owner, *a, **k)
        if capturing and owner is not None and not any(u is t for t in ops._PIN_SINK[0]):
            calls["n"] += 1          # a frozen layer's filters were transformed inside the capture
        return u

    ops._wino_weights = counting
    try:
        step = GraphedTrainStep(model, opt, batch)
    finally:
        ops._wino_weights = real
    assert calls["n"] == 0, f"{calls['n']} frozen-weight transforms were recorded into the training graph"
    assert len({t.data_ptr() for t in step._pins}) >= 16, "the capture was not served the cached VGG filters"
    r1 = float(step(batch).minimize)
    assert np.isfinite(r1)
    ops.invalidate_weight_caches()
    with pytest.raises(RuntimeError, match="captured"):
        step(batch)

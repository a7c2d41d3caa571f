"""Generates tests/golden/full/*.npz: the oracle (oracle/*.py) evaluated at the FULL-SIZE parity cases in fp32 and fp64
(plus the kink-shifted fp64 passes), in the build container, so that `pytest -m gpu` on the GPU box never runs a CPU
oracle at full size (~700 s of the 1200 s GPU run was this, with 2.5x host-to-host variance).

    python tests/golden/make_golden_fullsize.py            # every case (about 1.5 h on 8 cores, <= 45 GiB)
    python tests/golden/make_golden_fullsize.py chain_bs4 c5 ...

The oracle is pinned to the imported reference by tests/golden/make_golden.py / tests/test_oracle_golden.py.  Cases and their
inputs: tests/fullsize_cases.py; format: tests/gradfix.py.

Round 5: where /root/reference is present (the build container) the FP32 leg of the try-on / warp cases is the REFERENCE's
own evaluation - `models.warp_model.WarpModel` / `models.unet_mask_model.UnetMaskModel` imported through make_golden.py's
shim (torchvision / pytorch_lightning / tensorboard / flownet2 stand-ins), `training_step` + `backward` on the same procedural
weights and inputs - and the fixture says so (`fp32_source` = "reference").  The fp64 leg and the kink bracket stay the
oracle's (the reference has no fp64 mode worth trusting: its hard-coded `.float()` casts are few, but the oracle is the
audited restatement).  SHINEON_GOLDEN_FP32=oracle forces the round-4 behaviour (oracle in fp32).
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

import shineon_virtual_tryon_amd  # noqa: E402,F401
import fullsize_cases as fc  # noqa: E402
import gradfix as gf  # noqa: E402
import sams_helpers as sh  # noqa: E402
from oracle import sams_oracle as so  # noqa: E402
from oracle import shineon_oracle as oracle  # noqa: E402


REF = "/root/reference"
USE_REFERENCE = os.path.isdir(REF) and os.environ.get("SHINEON_GOLDEN_FP32", "reference") == "reference"
_SHIM = [False]


def _reference():
    """Import hook for the reference's modules (build container only)."""
    if not _SHIM[0]:
        sys.path.insert(0, HERE)
        import make_golden as mg   # the shim lives there; importing it has no side effects besides sys.path

        mg.install_shim()
        _SHIM[0] = mg
    return _SHIM[0]


def reference_warp(sd, batch):
    """models/warp_model.py:74-98 on the reference's own WarpModel: (gradients by state_dict key, outputs, BatchNorm updates)."""
    mg = _reference()
    from models.warp_model import WarpModel

    hp = mg.hp_namespace(person_inputs=fc.WHP["person_inputs"], cloth_inputs=fc.WHP["cloth_inputs"])

    def fresh():
        m = WarpModel(hp)
        m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        return m.train()

    model = fresh()
    res = model.training_step(dict(batch), 0)
    res.minimize.backward()
    with torch.no_grad():   # grid / theta of the same training-mode forward, from a twin (training_step does not keep them)
        grid, theta = fresh()(torch.cat([batch[k] for k in hp.person_inputs], 1), torch.cat([batch[k] for k in hp.cloth_inputs], 1))
    msd = model.state_dict()
    bn = {k: msd[k].detach().clone() for k in msd if k.endswith(("running_mean", "running_var"))}
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    ref = {"theta": theta, "grid": grid, "warped_cloth": model.warped_cloth.detach(), "loss/G": res.minimize.detach()}
    return grads, ref, bn


def reference_unet(sd, batch, hp):
    """models/unet_mask_model.py:137-217 on the reference's own UnetMaskModel (the VGG19 of the loss holds `sd`'s weights)."""
    mg = _reference()
    from models.unet_mask_model import UnetMaskModel

    ns = mg.hp_namespace(**{k: v for k, v in hp.items()})
    model = UnetMaskModel(ns)
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model.train()
    res = model.training_step(dict(batch), 0)
    res.minimize.backward()
    cat = lambda ts: None if ts is None or ts[0] is None else torch.cat([t.detach() for t in ts], 1)  # noqa: E731
    ref = {"p_rendereds": cat(model.p_rendereds), "tryon_masks": cat(model.tryon_masks), "p_tryons": cat(model.p_tryons),
           "flow_masks": cat(model.flow_masks) if getattr(model, "flow_masks", None) is not None else None}
    for k in fc.UNET_LOG_KEYS:
        ref[k] = res.logs[k].detach() if k in res.logs else torch.zeros(())
    grads = {"unet." + k: p.grad.detach().clone() for k, p in model.unet.named_parameters() if p.grad is not None}
    ref["_model"] = model
    return grads, ref


def _params(sd, trainable, dtype):
    return {k: (v.to(dtype).clone().requires_grad_(bool(trainable(k))) if v.is_floating_point() else v.clone())
            for k, v in sd.items()}


def _cast(d, dtype):
    return {k: (v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in d.items()}


def _grads(params):
    return {k: v.grad for k, v in params.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}


def oracle_warp(sd, batch, dtype=torch.float32, bn_updates=None):
    params = _params(sd, lambda k: "running" not in k and "num_batches" not in k, dtype)
    consts = _cast(oracle.tps_constants(256, 192, 5), dtype)
    ref = oracle.warp_losses(params, _cast(batch, dtype), fc.WHP, consts, bn_updates=bn_updates)
    ref["loss/G"].backward()
    return params, ref


def oracle_unet(sd, batch, hp, dtype=torch.float32):
    params = _params(sd, lambda k: k.startswith("unet."), dtype)
    ref = oracle.unet_mask_losses(params, _cast(batch, dtype), hp)
    ref["loss/G"].backward()
    return params, ref


def kink_spread(run64, exact):
    spread = {}
    for sign in (1.0, -1.0):
        with sh.kink_shift(sign * sh.KINK_DELTA):
            shifted = run64()
        for k, v in shifted.items():
            if k in exact:
                spread[k] = max(spread.get(k, 0.0), float((v - exact[k]).abs().max()))
    return spread


def gen_warp(bs, out=None, prefix=""):
    out = {} if out is None else out
    batch = fc.smooth_batch(bs)
    _, sd = fc.build_warp()
    bn = {}
    if USE_REFERENCE:
        g32, r32, bn = reference_warp(sd, batch)
    else:
        p32, r32 = oracle_warp(sd, batch, bn_updates=bn)
        g32 = _grads(p32)
    out[prefix + "fp32_source"] = np.array("reference" if USE_REFERENCE else "oracle")
    p64, r64 = oracle_warp(sd, batch, torch.float64)
    g64 = _grads(p64)
    assert set(g32) == set(g64), set(g32) ^ set(g64)
    kink = kink_spread(lambda: _grads(oracle_warp(sd, batch, torch.float64)[0]), g64)
    out[prefix + "digest:weights"] = gf.input_digest(sd)
    out[prefix + "digest:batch"] = gf.input_digest(batch)
    out[prefix + "theta32"] = r32["theta"].detach().numpy()
    out[prefix + "theta64"] = r64["theta"].detach().numpy()
    gf.pack_output(out, prefix + "grid", r32["grid"].permute(0, 3, 1, 2), r64["grid"].permute(0, 3, 1, 2))
    gf.pack_output(out, prefix + "warped_cloth", r32["warped_cloth"], r64["warped_cloth"])
    out[prefix + "handoff_f16"] = r32["warped_cloth"].detach().to(torch.float16).numpy()
    out[prefix + "loss32"] = np.float64(float(r32["loss/G"]))
    out[prefix + "loss64"] = np.float64(float(r64["loss/G"]))
    for k, v in bn.items():
        out[prefix + "bn:" + k] = v.detach().numpy()
    gf.pack_grads(out, prefix + "grad:", g32, g64, kink)
    return out, r32


def gen_unet(batch, sd, hp, out=None, prefix="", kinks=True):
    out = {} if out is None else out
    if USE_REFERENCE:
        g32, r32 = reference_unet(sd, batch, hp)
    else:
        p32, r32 = oracle_unet(sd, batch, hp)
        g32 = _grads(p32)
    out[prefix + "fp32_source"] = np.array("reference" if USE_REFERENCE else "oracle")
    p64, r64 = oracle_unet(sd, batch, hp, torch.float64)
    g64 = _grads(p64)
    assert set(g32) == set(g64), set(g32) ^ set(g64)
    # GELU U-Net: the only kinks are the VGG's ReLUs / max-pools (frozen weights; they shape dL/dp_tryon) and L1's sign
    kink = kink_spread(lambda: _grads(oracle_unet(sd, batch, hp, torch.float64)[0]), g64) if kinks else None
    out[prefix + "digest:weights"] = gf.input_digest(sd)
    out[prefix + "digest:batch"] = gf.input_digest(batch)
    for name in ("p_rendereds", "tryon_masks", "p_tryons", "flow_masks"):
        if r32[name] is not None:
            gf.pack_output(out, prefix + name, r32[name], r64[name])
    for k in fc.UNET_LOG_KEYS:
        out[prefix + "log32:" + k] = np.float64(float(r32[k]))
        out[prefix + "log64:" + k] = np.float64(float(r64[k]))
    gf.pack_grads(out, prefix + "grad:", g32, g64, kink)
    return out


def _chain(bs):
    out, r32 = gen_warp(bs, prefix="warp:")
    batch = dict(fc.smooth_batch(bs))
    batch["cloth"] = fc.handoff_cloth(r32["warped_cloth"])
    _, usd = fc.build_unet()
    return gen_unet(batch, usd, fc.UHP, out, prefix="tryon:")


def case_warp_bs2():
    """The inputs of tests/test_models_gpu.py::test_warp_model_vs_reference_golden (the REFERENCE's own bs=2 golden): the
    oracle's fp64 values beside the reference's fp32 ones, so that the test can say how far the reference itself is from exact."""
    out, _ = gen_warp(2)
    return out


def case_chain_bs4():
    return _chain(4)


def case_chain_bs8():
    return _chain(8)


def case_c5():
    _, sd = fc.build_c5()
    batch = fc.flatten_frames(fc.smooth_batch(1, n_frames=5))
    return gen_unet(batch, sd, fc.C5HP)


def case_c5_bs2():
    """BASELINE config 5 at the batch bench.py times it at (bs = 2 sequences of 5 frames): other GEMM M extents, other
    committed plans / tiles / split-K orders than the bs = 1 case, and the step goes through the graph-replayed TrainStep."""
    _, sd = fc.build_c5()
    batch = fc.flatten_frames(fc.smooth_batch(2, n_frames=5))
    return gen_unet(batch, sd, fc.C5HP)


def case_c3_bench_inputs():
    """Forward only, on the inputs bench.py TIMES: synthetic_batch(4, seed=420, smooth=False) - U(-1, 1) white-noise images
    (SURVEY 8d / BASELINE.md 3), not the band-limited images of the gradient cases.  UnetMaskModel outputs, the five logged
    scalars and the five VGG19 taps of p_tryon from the reference's own modules; fp64 from the oracle."""
    from shineon_virtual_tryon_amd.data import synthetic_batch

    assert USE_REFERENCE, "this case records the reference's own forward: run it in the build container"
    batch = synthetic_batch(4, "cpu", seed=420, smooth=False)
    _, sd = fc.build_unet()
    out = {"digest:weights": gf.input_digest(sd), "digest:batch": gf.input_digest(batch), "fp32_source": np.array("reference")}
    with torch.no_grad():
        r64 = oracle.unet_mask_losses(_params(sd, lambda k: False, torch.float64), _cast(batch, torch.float64), fc.UHP)
    _, r32 = reference_unet(sd, batch, fc.UHP)   # (its backward pass is simply not recorded)
    for name in ("p_rendereds", "tryon_masks", "p_tryons"):
        # p_tryon (what the loss sees) and the mask as FULL tensors: no sampling argument; p_rendered on a 1/4 lattice
        gf.pack_output(out, name, r32[name], r64[name], stride=2 if name == "p_rendereds" else 1)
    for k in fc.UNET_LOG_KEYS:
        out["log32:" + k] = np.float64(float(r32[k]))
        out["log64:" + k] = np.float64(float(r64[k]))
    model = r32["_model"]
    with torch.no_grad():
        taps32 = model.criterionVGG.vgg(r32["p_tryons"])
        x64 = r64["p_tryons"]
        taps64 = oracle.vgg19_features({k: v.double() for k, v in sd.items() if v.is_floating_point()}, x64)
    for i, (a, b) in enumerate(zip(taps32, taps64)):
        gf.pack_output(out, f"vgg_tap{i + 1}", a, b, stride=(8, 8, 4, 4, 2)[i])
    return out


def case_c2_bench_inputs():
    """Forward only, on the inputs bench.py TIMES for the warp stage (c2 / c4): synthetic_batch(4, seed=420, smooth=False).
    theta (full), the TPS grid (as (B, 2, H, W), 1/2 lattice + whole-tensor checksums) and the loss of the REFERENCE's own
    WarpModel in training mode (models/warp_model.py:63-98); fp64 from the oracle.  The warped cloth is not recorded: on
    white-noise cloth every bilinear tap flip is an O(1) difference (VERDICT r05 item 8 asks for theta and the grid)."""
    from shineon_virtual_tryon_amd.data import synthetic_batch

    assert USE_REFERENCE, "this case records the reference's own forward: run it in the build container"
    batch = synthetic_batch(4, "cpu", seed=420, smooth=False)
    _, sd = fc.build_warp()
    out = {"digest:weights": gf.input_digest(sd), "digest:batch": gf.input_digest(batch), "fp32_source": np.array("reference")}
    _, r32, _ = reference_warp(sd, batch)
    _, r64 = oracle_warp(sd, batch, torch.float64)
    b = r32["theta"].shape[0]
    gf.pack_output(out, "theta", r32["theta"].reshape(b, 1, 1, -1), r64["theta"].reshape(b, 1, 1, -1), stride=1)
    gf.pack_output(out, "grid", r32["grid"].permute(0, 3, 1, 2), r64["grid"].permute(0, 3, 1, 2), stride=2)
    out["log32:loss/G"] = np.float64(float(r32["loss/G"]))
    out["log64:loss/G"] = np.float64(float(r64["loss/G"]))
    return out


# ---- SAMS --------------------------------------------------------------------------------------------
def reference_sams_three_steps(sd, hp, batch):
    """models/sams_model.py:147-383 on the reference's own SamsModel, in Lightning 0.9's multi-optimizer order (only the
    current optimizer's parameters require grad): the structure sams_helpers.oracle_three_steps returns - per step (logs,
    gradients by state_dict key), the generated frames of step 0, the state_dict after the three steps."""
    mg = _reference()
    from models.sams_model import SamsModel

    ns = mg.hp_namespace(**vars(hp))
    torch.manual_seed(0)
    model = SamsModel(ns)
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model.train()
    nets = {0: ("generator", model.generator), 1: ("multiscale_discriminator", model.multiscale_discriminator),
            2: ("temporal_discriminator", model.temporal_discriminator)}
    out, frames = [], None
    for idx in (0, 1, 2):
        own, net = nets[idx]
        for q in model.parameters():
            q.requires_grad = False
        for q in net.parameters():
            q.requires_grad = True
        model.zero_grad()
        res = model.training_step(dict(batch), 0, idx)
        res.minimize.sum().backward()
        if idx == 0:
            frames = model.all_gen_frames.detach().clone()
        out.append(({k: float(v.detach().sum()) for k, v in res.logs.items()},
                    {f"{own}.{k}": q.grad.detach().clone() for k, q in net.named_parameters() if q.grad is not None}))
    return out, frames, {k: v.detach().clone() for k, v in model.state_dict().items()}


def _sams_three_steps(sd, hp, batch, out, fp64_kinks=True, o97=False):
    if USE_REFERENCE:   # round 6: the fp32 leg is the reference's own SamsModel (VERDICT r05 item 8 ii)
        ref32, frames32, sd32 = reference_sams_three_steps(sd, hp, batch)
        oref, _, _ = sh.oracle_three_steps(sd, hp, batch) if o97 else (None, None, None)
        out["fp32_source"] = np.array("reference")
    else:
        ref32, frames32, sd32 = sh.oracle_three_steps(sd, hp, batch)
        oref = ref32
        out["fp32_source"] = np.array("oracle")
    ref64, frames64, sd64 = sh.oracle_three_steps(sd, hp, batch, torch.float64)
    kinks = sh.kink_spread(sd, hp, batch, ref64) if fp64_kinks else [None] * 3
    out["digest:weights"] = gf.input_digest(sd)
    out["digest:batch"] = gf.input_digest(batch)
    for idx in range(3):
        for k in ref32[idx][0]:
            out[f"log32:{idx}:{k}"] = np.float64(ref32[idx][0][k])
            out[f"log64:{idx}:{k}"] = np.float64(ref64[idx][0][k])
        gf.pack_grads(out, f"grad{idx}:", ref32[idx][1], ref64[idx][1], kinks[idx])
        if o97:  # the fp32 ORACLE at the sample positions of the REFERENCE goldens (tests/golden/sams_*.npz, gs<idx>: keys)
            for k, v in oref[idx][1].items():
                out[f"o97:{idx}:{k}"] = v.detach().contiguous().reshape(-1)[::97].float().numpy()
    f32 = frames32.reshape(frames32.shape[0], -1, *frames32.shape[-2:])
    f64 = frames64.reshape(frames64.shape[0], -1, *frames64.shape[-2:])
    gf.pack_output(out, "frames", f32, f64, stride=3 if f32.shape[-1] > 64 else 1)
    out["frames:max64"] = np.float64(frames64.abs().max().item())
    # buffers after the three steps (power-iteration vectors, running statistics, counters)
    names, vals = [], []
    for k, v in sd64.items():
        if k.startswith("criterion_VGG"):
            continue
        if k.endswith(("weight_u", "weight_v", "running_mean", "running_var")):
            out["buf:" + k] = v.detach().double().numpy()
        if k.endswith("num_batches_tracked"):
            names.append(k)
            vals.append(int(sd32[k]))
    out["nbt:names"] = np.array(names) if names else np.array([], dtype="<U1")
    out["nbt:values"] = np.array(vals, np.int64)
    return out


def _case_sams_small(tag):
    _, sd, hp, batch = fc.sams_small_case(tag)
    out = _sams_three_steps(sd, hp, batch, {}, o97=True)
    if tag == "base":
        _sams_two_iterations(sd, hp, batch, out)
    return out


def _sams_two_iterations(sd, hp, batch, out):
    """Two full iterations (three Adam optimizers, Lightning's order) of the fp64 oracle: every logged scalar of both
    iterations and the total update of every parameter (tests/test_sams_gpu.py::test_sams_two_full_iterations...)."""
    osd = {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    obatch = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    groups = so.optimizer_groups(osd)
    model = so.SamsOracle(osd, hp)
    opts = []
    for net, lr in zip(sh.STEP_NETS, (hp.lr, hp.lr_D, hp.lr_D)):
        for k in groups[net]:
            osd[k].requires_grad_(True)
        opts.append(torch.optim.Adam([osd[k] for k in groups[net]], lr))
    for it in range(2):
        for idx, net in enumerate(sh.STEP_NETS):
            for k, v in osd.items():
                if v.is_floating_point():
                    v.requires_grad_(k in groups[net])
            fn = (model.generator_step, model.multiscale_discriminator_step, model.temporal_discriminator_step)[idx]
            loss, logs = fn(obatch)
            opts[idx].zero_grad()
            loss.sum().backward()
            opts[idx].step()
            for k, v in logs.items():
                out[f"adam:log:{it}:{idx}:{k}"] = np.float64(float(v.detach().sum()))
    for net in sh.STEP_NETS:
        for k in groups[net]:
            # in units of 1e-4 (the generator's learning rate) as fp16: Adam's first steps move every element by ~lr, and the
            # test compares direction (cosine) and length of the whole tensor's update, not elements
            out["adam:update_e4_f16:" + k] = ((osd[k].detach() - sd[k].double()) * 1e4).to(torch.float16).numpy()


def case_sams_base():
    return _case_sams_small("base")


def case_sams_attn_gelu():
    return _case_sams_small("attn_gelu")


def case_sams_progressive():
    return _case_sams_small("progressive")


def case_sams_full_three_steps():
    hp, _, sd, batch = fc.sams_full_three_steps_case()
    return _sams_three_steps(sd, hp, batch, {})


def case_sams_full_generator_bs4():
    hp, _, sd, prev_frames, prev_maps, maps, gout = fc.sams_full_generator_case()

    def run(dtype):
        osd = {k: (v.to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for k, v in osd.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var", "weight_u", "weight_v")):
                v.requires_grad_(True)
        o = so.generator_forward(osd, prev_frames.to(dtype), prev_maps.to(dtype), {k: v.to(dtype) for k, v in maps.items()}, hp, True)
        o.backward(gout.to(dtype))
        return o.detach(), {k[len("generator."):]: v.grad for k, v in osd.items() if v.requires_grad and v.grad is not None}

    def run_reference():
        """models/networks/sams/sams_generator.py:240-291 on the reference's own SamsGenerator (training mode)."""
        mg = _reference()
        from models.networks.sams.sams_generator import SamsGenerator

        torch.manual_seed(0)
        gen = SamsGenerator(mg.hp_namespace(**vars(hp)))
        gen.load_state_dict({k[len("generator."):]: v.clone() for k, v in sd.items()}, strict=True)
        gen.train()
        o = gen(prev_frames.clone(), prev_maps.clone(), {k: v.clone() for k, v in maps.items()})
        o.backward(gout)
        return o.detach(), {k: q.grad.detach().clone() for k, q in gen.named_parameters() if q.grad is not None}

    o32o, g32o = run(torch.float32)
    o32, g32 = run_reference() if USE_REFERENCE else (o32o, g32o)
    o64, g64 = run(torch.float64)
    spread = {}
    for sign in (1.0, -1.0):  # as the r03 test: the bracket from two more FP32 passes at this size (oracle against oracle)
        with sh.kink_shift(sign * sh.KINK_DELTA):
            _, g = run(torch.float32)
        for k, v in g.items():
            spread[k] = max(spread.get(k, 0.0), (v - g32o[k]).abs().max().item())
    out = {"digest:weights": gf.input_digest(sd), "fp32_source": np.array("reference" if USE_REFERENCE else "oracle"),
           "digest:inputs": gf.input_digest({"pf": prev_frames, "pm": prev_maps, "gout": gout, **maps})}
    gf.pack_output(out, "out", o32, o64)
    out["out:max64"] = np.float64(o64.abs().max().item())
    gf.pack_grads(out, "grad:", g32, g64, spread)
    return out


CASES = {
    "warp_bs2": case_warp_bs2, "chain_bs4": case_chain_bs4, "chain_bs8": case_chain_bs8,
    "c5": case_c5, "c5_bs2": case_c5_bs2, "c3_bench_inputs": case_c3_bench_inputs, "c2_bench_inputs": case_c2_bench_inputs, "sams_base": case_sams_base, "sams_attn_gelu": case_sams_attn_gelu,
    "sams_progressive": case_sams_progressive, "sams_full_generator_bs4": case_sams_full_generator_bs4,
    "sams_full_three_steps": case_sams_full_three_steps,
}


def main(argv):
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    for name in (argv or list(CASES)):
        t0 = time.time()
        out = CASES[name]()
        path = gf.save(name, out)
        print(f"{name}: {os.path.getsize(path) / 2 ** 20:.2f} MiB, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])

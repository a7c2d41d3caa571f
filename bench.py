"""Headline benchmark: try-on frames/sec (forward+backward+Adam) at 256x192, bs=4 per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

--config c4 (default, the headline): a "step" is the chained warp -> try-on training step of SURVEY.md §8d (C4 at the
BASELINE batch size): WarpModel (GMM) training step, then UnetMaskModel (self-attention, num_attn=2, GELU, L1 + VGG +
mask loss) training step on the detached warped cloth, each with its Adam update.
--config c2 / c3: BASELINE config 2 (WarpModel alone) / config 3 (UnetMaskModel alone), same protocol.
Under N > 1 each rank works on its own frames (weak scaling) and gradients are mean-all-reduced over RCCL.
Inputs are synthetic (seed 420) and resident in HBM before the timed region; weights are random-init.

The step that is timed is the PRODUCT's step: shineon_virtual_tryon_amd.trainer.ChainedTrainStep / TrainStep, the same
engines Trainer.fit drives.  Rank 0 prints ONE JSON line.  `roofline` carries the live HIP-event measurement of the
dominant MFMA kernel, the step-level fraction (algorithmic FLOPs of the whole step / step time / fp32-MFMA peak) and an
`hbm` table (GB/s vs 8 TB/s) for the bandwidth-bound kernels; `cpu_baseline` is the oracle (a CPU restatement of the
reference, kind "port") timed on this host's cores: 2 warm-up + >= 5 timed steps, median.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import shineon_virtual_tryon_amd as pkg  # noqa: E402
from shineon_virtual_tryon_amd.data import synthetic_batch  # noqa: E402
from shineon_virtual_tryon_amd import trainer as so_trainer  # noqa: E402
from shineon_virtual_tryon_amd.trainer import Trainer, TrainStep, broadcast_parameters  # noqa: E402
from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel  # noqa: E402
from shineon_virtual_tryon_amd.warp_model import WarpModel  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
NKEYS = 40
KEY_NAMES = [f"{m}_{t}" for m in ("fprop", "dgrad", "wgrad", "gemm", "winograd_gemm")
             for t in ("64x64", "128x64", "64x128", "128x128", "128x128w8", "64x128w8", "thin4", "-")]
KEY_NAMES[7] = "winograd_fused"   # csrc/wino.hip: FLOPs recorded = algorithmic (direct-convolution) FLOPs; executed = / 2.25
KEY_NAMES[39] = "winograd_pgemm"  # csrc/pgemm.hip: the Winograd-domain (F(4x4)) products that run in the persistent NT GEMM
WINOGRAD_KEY, WINOGRAD_FACTOR, PGEMM_KEY = 7, 2.25, 39
# algorithmic GFLOP per frame: GMM fwd+bwd 28.0; try-on step = U-Net 50.3 + VGG19 (2 fwd + 1 dgrad) 106.5
GF_PER_FRAME = {"c1": 16.8, "c2": 28.0, "c3": 156.8, "c4": 184.8, "c5": None, "sams": 3131.4}  # None: the MFMA launches' own 2MNK sum
# sams: 62 627 GF per bs = 4 x 5-frame step = the sum of 2MNK over every MFMA launch of the step as DIRECT convolutions
# (profiles/r02_sams_bench_bs4.json; 15 657 GF at bs = 1: linear in the batch).  Since round 3 part of the 3x3 convolutions
# run as Winograd F(2x2,3x3) and execute fewer FLOPs; the algorithmic figure stays the direct-convolution one.
WORKLOADS = {
    "c1": "BASELINE config 1: UnetMaskModel FORWARD (self_attn, num_attn=2, gelu; U-Net + tanh/sigmoid/mask blend) on 4x256x192 "
          "random tensors, no backward, no loss",
    "c4": "chained warp->try-on training step (SURVEY 8d C4 at bs=4/GPU): WarpModel (GMM) fwd+bwd+Adam, then UnetMaskModel "
          "(self_attn, num_attn=2, gelu; L1+VGG19+mask loss) fwd+bwd+Adam on the warped cloth, 256x192",
    "c2": "BASELINE config 2: WarpModel (GMM feature-extract + correlation + TPS grid_sample) fwd+bwd+Adam, 256x192",
    "c3": "BASELINE config 3: UnetMaskModel (self_attn, num_attn=2, gelu; L1+VGG19+mask loss) fwd+bwd+Adam, 256x192",
    "c5": "BASELINE config 5: UnetMaskModel with n_frames_total=5, flow_warp (ngf=167, 50 in / 25 out channels, 154 M parameters, "
          "Resample2d chain, flow-mask penalty), bs=2 sequences of 5 frames per GPU, fwd+bwd+Adam, 256x192; flow fields synthetic "
          "(FlowNet2 stays upstream)",
    "sams": "SURVEY 8f-4: SamsModel (Self-Attentive Multi-SPADE GAN) with the reference's default networks (generator 64..1024 "
            "features, 4 encoder + 3 middle + 4 decoder SPADE blocks, spectral norm; 2-scale + temporal PatchGAN discriminators, "
            "hinge loss), n_frames_total=5, flow_warp, 256x192; one step = the three optimizers in Lightning's order "
            "(generator: 5 generator passes fwd+bwd + both discriminators + L1 + VGG19; multiscale D: 5 more generator passes "
            "under no_grad + D fwd+bwd; temporal D fwd+bwd), each with its Adam update",
}


def sams_hparams(**kw):
    base = dict(n_frames_total=5, n_frames_now=None, person_inputs=["agnostic", "densepose", "flow"], cloth_inputs=["cloth"],
                encoder_input="flow", flow_warp=True, activation="relu", fine_height=256, fine_width=192, is_train=True,
                norm_G="spectralspadesyncbatch3x3", ngf_base=2, ngf_pow_outer=6, ngf_pow_inner=10, ngf_pow_step=1, num_middle=3,
                attention_middle_indices=[], attention_decoder_indices=[], init_type="xavier", init_variance=0.02,
                netD_subarch="n_layer", num_D=2, n_layers_D=4, ndf=64, norm_D="spectralinstance", gan_mode="hinge", lr=1e-4,
                lr_D=3e-4, no_ganFeat_loss=False, wt_l1=1.0, wt_vgg=1.0, wt_multiscale=1.0, wt_temporal=1.0,
                display_count=10 ** 9, keep_epochs=5, decay_epochs=5, allow_random_vgg=True)
    base.update(kw)
    return argparse.Namespace(**base)


def sams_cpu_baseline(batch_size):
    """The SAMS oracle on this host's cores, on a BOUNDED sample of the step: the same networks at 256x192 with
    n_frames_total=2 (one previous frame instead of four: 2/5 of the generator passes, everything else per step unchanged),
    one untimed + one timed three-optimizer step."""
    from oracle import sams_oracle as so
    from oracle.procedural import procedural_state_dict, shapes_of
    from shineon_virtual_tryon_amd.sams_model import SamsModel

    torch.set_num_threads(usable_cores())
    hp = sams_hparams(n_frames_total=2)
    sd = procedural_state_dict(shapes_of(SamsModel(hp).state_dict()))
    batch = synthetic_batch(batch_size, "cpu", n_frames=2)
    groups = so.optimizer_groups(sd)
    oracle = so.SamsOracle(sd, hp)
    nets = ("generator", "multiscale_discriminator", "temporal_discriminator")
    for net in nets:
        for k in groups[net]:
            sd[k].requires_grad_(True)
    opts = [torch.optim.Adam([sd[k] for k in groups[net]], lr) for net, lr in zip(nets, (hp.lr, hp.lr_D, hp.lr_D))]

    def step():
        for idx, net in enumerate(nets):
            for k, v in sd.items():
                if v.is_floating_point():
                    v.requires_grad_(k in groups[net])
            loss, _ = (oracle.generator_step, oracle.multiscale_discriminator_step, oracle.temporal_discriminator_step)[idx](batch)
            opts[idx].zero_grad()
            loss.sum().backward()
            opts[idx].step()

    step()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    dt = times[1]
    return {"value": batch_size * 2 / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "cpu": cpu_model(),
            "kind": "port (bounded sample)",
            "sample": f"1 warm-up + 3 timed three-optimizer SamsModel steps of the oracle at 256x192, bs={batch_size}, BOUNDED to "
                      f"n_frames_total=2 (2 instead of 5 generator passes per generation: the full step would take minutes), "
                      f"PyTorch CPU fp32: median {dt:.1f} s/step (min {times[0]:.1f}, max {times[2]:.1f})"}


def ramp_clocks(step, seconds):
    """Untimed: the step itself, repeated for about `seconds` of wall time before the warm-up steps.  The timed window of the
    short configurations is 0.02 s (c1) to 0.13 s (c4); started right after the capture / CPU-side set-up it sometimes began on
    the clock ramp out of idle (profiles/README.md: c1 1653 instead of 3930 frames/s, c3 615 instead of 800, same build).  The
    sustained runs under profiles/ (>= 2 s) never showed it.  With several ranks the steps contain collectives, so every rank
    runs the SAME number of steps: two are timed, the slowest rank's time fixes the count.  Returns the steps run."""
    if seconds <= 0:
        return 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    step()
    torch.cuda.synchronize()
    per_step = (time.perf_counter() - t0) / 2
    if so_trainer._collective():
        t = torch.tensor([per_step], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_step = float(t.item())
    n = max(0, min(20000, int(seconds / max(per_step, 1e-5)) - 2))
    for i in range(n):
        step()
        if i % 4 == 3:
            torch.cuda.synchronize()   # the host must not queue seconds of work ahead of the device
    torch.cuda.synchronize()
    return n + 2


def dominant_roofline(ms, fl, by, cnt, traffic, traffic_source, timing):
    """The `roofline` entries of the kernel with the largest summed time.  `achieved` / `frac` are what the matrix pipe
    EXECUTES (<= 1 by construction - asserted); the direct-convolution (algorithmic) figure, which a Winograd kernel exceeds
    by 2.25x, is kept as algorithmic_tflops / algorithmic_frac.  traffic_ratio = PMC HBM bytes per launch / algorithmic bytes
    per launch (operands once + result once, summed by the library's profiler over this kernel's launches)."""
    dom = max(range(NKEYS), key=lambda k: ms[k])
    algorithmic = fl[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
    executed = algorithmic / WINOGRAD_FACTOR if dom == WINOGRAD_KEY else algorithmic
    frac = executed / PEAK_FP32_MFMA_TFLOPS
    assert 0.0 <= frac <= 1.0, f"roofline.frac {frac} is not a fraction: executed {executed} TFLOP/s vs peak {PEAK_FP32_MFMA_TFLOPS}"
    alg_bytes = by[dom] / cnt[dom] if cnt[dom] and by[dom] > 0 else None
    return dom, {
        "bound": "mfma",
        "kernel": ("wino_fused_k (Winograd F(2x2,3x3), csrc/wino.hip)" if dom == WINOGRAD_KEY else
                   "pgemm_nt_k (persistent NT GEMM of the Winograd F(4x4,3x3)-domain products, csrc/pgemm.hip)" if dom == PGEMM_KEY else
                   f"so_igemm_kernel<{KEY_NAMES[dom]}>"),
        "achieved": executed, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": frac,
        "achieved_note": ("EXECUTED MFMA FLOPs / kernel time" + (": the Winograd F(2x2,3x3) kernel executes 1/2.25 of the "
                          "direct-convolution multiplications; algorithmic_* is the direct-convolution figure (2 x pixels x Ko x 9C, "
                          "SURVEY 8d) over the same time and may exceed the peak of a direct kernel" if dom == WINOGRAD_KEY else
                          " (= algorithmic FLOPs for this kernel)")),
        "algorithmic_tflops": algorithmic, "algorithmic_frac": algorithmic / PEAK_FP32_MFMA_TFLOPS,
        "executed_tflops": executed, "executed_frac": frac,
        "traffic": traffic, "traffic_source": traffic_source if traffic else None,
        "algorithmic_bytes": alg_bytes,
        "traffic_ratio": (traffic / alg_bytes) if traffic and alg_bytes else None,
        "timing": timing, "avg_launch_us": 1e3 * ms[dom] / max(1, cnt[dom]),
    }


def sams_traffic(key):
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None
    return json.load(open(tpath)).get("sams", {}).get(key, {}).get("hbm_bytes_per_launch")


def sams_traffic_source():
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    src = (json.load(open(tpath)).get("sams", {}).get("_sources") or [""]) if os.path.exists(tpath) else [""]
    if "sams_step" in src[0]:
        return ("profiles/traffic.json [sams]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on THIS step (bs = 4, one step, counters "
                "restricted to the kernel with --kernel-include-regex), bytes per launch averaged over its launches")
    return ("profiles/traffic.json [sams]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on the kernel's most frequent layer of this step "
            "(128 -> 256 channels, 256x192, bs = 4) in a single-layer process (tools/one_layer.py), bytes per launch")


def sams_hbm_table(dev, batch_size):
    """GB/s of the SAMS-only bandwidth-bound kernels (csrc/sams.hip) at the step's own tensor sizes: stand-alone launches,
    20 repetitions, HIP events on the launch stream, algorithmic bytes per DESIGN.md 3.6, against the 8 TB/s HBM peak."""
    from shineon_virtual_tryon_amd import ops_sams

    L = pkg.lib()
    st = torch.cuda.current_stream().cuda_stream
    f = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    n = batch_size
    rows, c = n * 256 * 192, 64
    x, gb, y, dy = f(rows, c), f(rows, 2 * c), f(rows, c), f(rows, c)
    dn, dgb = torch.empty_like(x), torch.empty_like(gb)
    part = torch.empty(L.so_spade_bwd_colsum_blocks(rows, c), 2 * c, device=dev)  # per-block column sums (bias gradient)
    cases = {}
    cases["spade modulation fwd + LeakyReLU (256x192, C=64; 16 B/elem)"] = (
        lambda: L.so_spade_fwd(x.data_ptr(), c, gb.data_ptr(), 2 * c, gb.data_ptr() + 4 * c, 2 * c, y.data_ptr(), c, rows, c, 2, 0.2, st),
        rows * c * 16)
    cases["spade modulation bwd + bias-gradient partials (256x192, C=64; 28 B/elem)"] = (
        lambda: L.so_spade_bwd(x.data_ptr(), c, gb.data_ptr(), 2 * c, gb.data_ptr() + 4 * c, 2 * c, dy.data_ptr(), c, dn.data_ptr(), c,
                               dgb.data_ptr(), 2 * c, dgb.data_ptr() + 4 * c, 2 * c, rows, c, 2, 0.2, part.data_ptr(), st), rows * c * 28)
    cases["residual add (256x192, C=64; 12 B/elem)"] = (
        lambda: L.so_add(x.data_ptr(), c, dy.data_ptr(), c, y.data_ptr(), c, rows, c, st), rows * c * 12)
    lab = f(n * 256 * 192, 4)
    lab2 = torch.empty(n * 128 * 96, 4, device=dev)
    cases["nearest resize of a label map (256x192 -> 128x96, C=4)"] = (
        lambda: L.so_resize_nearest_fwd(lab.data_ptr(), 4, lab2.data_ptr(), 4, n, 256, 192, 128, 96, 4, 2.0, 2.0, st), lab2.numel() * 8)
    xa = f(n * 128 * 96, 128)
    xb = torch.empty(n * 256 * 192, 128, device=dev)
    cases["nearest x2 of an activation (128x96 -> 256x192, C=128)"] = (
        lambda: L.so_resize_nearest_fwd(xa.data_ptr(), 128, xb.data_ptr(), 128, n, 128, 96, 256, 192, 128, 0.5, 0.5, st),
        (xa.numel() + xb.numel()) * 4)
    din = f(2 * n * 256 * 192, 15)
    dout = torch.empty(2 * n * 128 * 96, 15, device=dev)
    cases["avg_pool 3x3 s2 of the discriminator input (2B x 256x192, C=15)"] = (
        lambda: L.so_avgpool3s2_fwd(din.data_ptr(), 15, dout.data_ptr(), 15, 2 * n, 256, 192, 15, st), (din.numel() + dout.numel()) * 4)
    o, i, rs = 1024, 1024, 9
    w = f(o, rs, i) * 0.02
    wo = torch.empty_like(w)
    u = torch.nn.functional.normalize(f(o), dim=0)
    v = torch.nn.functional.normalize(f(i * rs), dim=0)
    sig = torch.empty(1, device=dev)
    ws = torch.empty(L.so_spectral_norm_ws_floats(o, i, rs), device=dev)
    cases["spectral norm fwd, 1024x1024x3x3 (3 reads + 1 write of W)"] = (
        lambda: L.so_spectral_norm_fwd(w.data_ptr(), o, i, rs, u.data_ptr(), v.data_ptr(), wo.data_ptr(), sig.data_ptr(), 1, 1e-12,
                                       ws.data_ptr(), st), w.numel() * 16)
    gw = f(o, rs, i)
    dw = torch.empty_like(w)
    cases["spectral norm bwd, 1024x1024x3x3 (3 reads + 1 write of W)"] = (
        lambda: L.so_spectral_norm_bwd(gw.data_ptr(), w.data_ptr(), u.data_ptr(), v.data_ptr(), sig.data_ptr(), o, i, rs, dw.data_ptr(), 0,
                                       ws.data_ptr(), st), w.numel() * 16)
    out = {}
    for name, (fn, nbytes) in cases.items():
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        gbs = nbytes / (us * 1e-6) / 1e9
        out[name] = {"bytes": nbytes, "avg_us": round(us, 2), "GB/s": round(gbs, 1), "frac_of_peak": round(gbs / PEAK_HBM_GBS, 3)}
    return out


def run_sams(args, trainer, L):
    """--config sams: SamsModel's three-optimizer step through trainer.MultiOptimizerStep (eager launches)."""
    from shineon_virtual_tryon_amd.sams_model import SamsModel
    from shineon_virtual_tryon_amd.trainer import MultiOptimizerStep

    rank, world, dev = trainer.rank, trainer.world, trainer.device
    hp = sams_hparams()
    nfr = hp.n_frames_total
    torch.manual_seed(420)
    model = SamsModel(hp).to(dev).train()
    model.global_step = 1
    batch = synthetic_batch(args.batch, dev, seed=420, start=rank * args.batch, n_frames=nfr)
    opts, _ = model.configure_optimizers()
    for o in opts:
        broadcast_parameters(model, optimizer=o)
    engine = MultiOptimizerStep(model, opts)
    nparams = [sum(p.numel() for p in net.parameters()) for net in model.optimizer_networks()]
    log(f"SamsModel: generator {nparams[0] / 1e6:.1f} M, multiscale D {nparams[1] / 1e6:.1f} M, temporal D {nparams[2] / 1e6:.1f} M "
        f"parameters; bs={args.batch} x {nfr} frames")
    for i in range(args.warmup):
        t_w = time.perf_counter()
        engine(batch, i)
        torch.cuda.synchronize()
        log(f"warm-up step {i}: {1e3 * (time.perf_counter() - t_w):.1f} ms, peak HBM {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")

    def fence():
        torch.cuda.synchronize()
        if so_trainer._collective():
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    L.so_prof_enable(1)
    t0 = time.perf_counter()
    for i in range(args.steps):
        engine(batch, i)
    fence()
    elapsed = time.perf_counter() - t0
    L.so_prof_enable(0)
    if so_trainer._collective():
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = (ctypes.c_float * NKEYS)()
    fl = (ctypes.c_float * NKEYS)()
    cnt = (ctypes.c_int * NKEYS)()
    by = (ctypes.c_double * NKEYS)()
    L.so_prof_collect_bytes(ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(cnt), ctypes.addressof(by))
    if rank != 0:
        return
    kernels = {KEY_NAMES[k]: {"launches": cnt[k], "avg_us": 1e3 * ms[k] / cnt[k], "total_ms_per_step": ms[k] / args.steps,
                              "tflops": fl[k] / (ms[k] * 1e-3) / 1e12} for k in range(NKEYS) if cnt[k] > 0}
    dom0 = max(range(NKEYS), key=lambda k: ms[k])
    dom, roof = dominant_roofline(
        ms, fl, by, cnt, sams_traffic(KEY_NAMES[dom0]) if args.batch == 4 else None,   # (the PMC passes ran at bs = 4)
        sams_traffic_source(), "hip events, eager launches in the timed region")
    for k_ in kernels.values():
        k_["executed_tflops"] = k_["tflops"]
    if "winograd_fused" in kernels:
        kernels["winograd_fused"]["executed_tflops"] = kernels["winograd_fused"]["tflops"] / WINOGRAD_FACTOR
    step_ms = 1e3 * elapsed / args.steps
    gf_step = GF_PER_FRAME["sams"] * args.batch * nfr
    mfma_ms = sum(ms) / args.steps
    out = {
        "metric": "SAMS-GAN video frames/sec (three-optimizer step, fwd+bwd+Adam) at 256x192, n_frames=5",
        "value": world * args.batch * nfr * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "n_ranks_seen": dist.get_world_size() if dist.is_initialized() else 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp32", "data": "synthetic",
        "config": {"workload": WORKLOADS["sams"], "config": "sams", "launch": "eager (hip events bracket every MFMA launch)",
                   "batch_per_gpu": args.batch, "global_batch": world * args.batch, "frames_per_sample": nfr,
                   "parallelism": f"dp{world}", "step_api": "shineon_virtual_tryon_amd.trainer.MultiOptimizerStep",
                   "parameters_M": [round(n / 1e6, 2) for n in nparams],
                   "peak_hbm_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
        "roofline": {
            **roof,
            "step": {"algorithmic_gflop_per_step": gf_step, "achieved": gf_step / step_ms,
                     "frac": gf_step / step_ms / PEAK_FP32_MFMA_TFLOPS,
                     "note": "direct-convolution FLOPs of the step (sum of 2MNK over its MFMA launches before Winograd, "
                             "profiles/r02_sams_bench_bs4.json) / measured step time / fp32-MFMA peak"},
            "mfma_ms_per_step": mfma_ms, "mfma_time_frac_of_step": mfma_ms / step_ms,
            "all_mfma_tflops": sum(fl) / (sum(ms) * 1e-3) / 1e12 if sum(ms) > 0 else 0.0,
        },
        "kernels": kernels,
    }
    if not args.no_hbm_table:
        del model, engine, opts, batch
        torch.cuda.empty_cache()
        out["roofline"]["hbm"] = {"peak_GB/s": PEAK_HBM_GBS, "kernels": sams_hbm_table(dev, args.batch)}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = sams_cpu_baseline(args.batch)
    print(json.dumps(flatten_roofline(out)), flush=True)



def run_c1(args, trainer, L):
    """--config c1 (BASELINE config 1): UnetMaskModel.forward alone - the reference's own CPU-runnable case - on the GPU (one
    hipGraph of the forward pass, replayed) with the oracle's forward timed beside it on this host's cores."""
    from oracle import shineon_oracle as oracle
    from oracle.procedural import procedural_state_dict, shapes_of

    rank, world, dev = trainer.rank, trainer.world, trainer.device
    model = UnetMaskModel(hparams(person_inputs=["agnostic", "densepose"])).to(dev).train()
    batch = synthetic_batch(args.batch, dev, seed=420, start=rank * args.batch)
    person = torch.cat([batch["agnostic"], batch["densepose"]], 1)
    cloth = batch["cloth"]
    with torch.no_grad():
        for _ in range(2):
            out = model(person, cloth)   # plans measured, scratch sized
        torch.cuda.synchronize()
        graph = None
        if not args.no_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model(person, cloth)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = model(person, cloth)

        def step():
            if graph is not None:
                graph.replay()
            else:
                model(person, cloth)

        def fence():
            torch.cuda.synchronize()
            if so_trainer._collective():
                dist.barrier()
            torch.cuda.synchronize()

        log(f"clock ramp: {ramp_clocks(step, args.ramp_seconds)} untimed steps in {args.ramp_seconds:.1f} s")
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        if so_trainer._collective():
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        prof_steps = min(args.steps, 5)
        L.so_prof_enable(1)
        for _ in range(prof_steps):
            model(person, cloth)
        fence()
        L.so_prof_enable(0)
    ms = (ctypes.c_float * NKEYS)()
    fl = (ctypes.c_float * NKEYS)()
    cnt = (ctypes.c_int * NKEYS)()
    by = (ctypes.c_double * NKEYS)()
    L.so_prof_collect_bytes(ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(cnt), ctypes.addressof(by))
    if rank != 0:
        return
    kernels = {KEY_NAMES[k]: {"launches": cnt[k], "avg_us": 1e3 * ms[k] / cnt[k], "total_ms_per_step": ms[k] / prof_steps,
                              "tflops": fl[k] / (ms[k] * 1e-3) / 1e12} for k in range(NKEYS) if cnt[k] > 0}
    dom, roof = dominant_roofline(ms, fl, by, cnt, None, None,
                                  f"hip events on the same kernels launched eagerly for {prof_steps} passes right after the timed region")
    step_ms = 1e3 * elapsed / args.steps
    gf_step = GF_PER_FRAME["c1"] * args.batch
    out = {
        "metric": "try-on frames/sec (UnetMaskModel forward only) at 256x192 bs=4",
        "value": world * args.batch * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "n_ranks_seen": dist.get_world_size() if dist.is_initialized() else 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp32", "data": "synthetic",
        "config": {"workload": WORKLOADS["c1"], "config": "c1", "launch": "eager" if graph is None else "one hipGraph (forward)",
                   "untimed_clock_ramp_s": args.ramp_seconds,
                   "batch_per_gpu": args.batch, "global_batch": world * args.batch, "frames_per_sample": 1, "parallelism": f"dp{world}",
                   "step_api": "shineon_virtual_tryon_amd.unet_mask_model.UnetMaskModel.forward"},
        "roofline": {**roof,
                     "step": {"algorithmic_gflop_per_step": gf_step, "achieved": gf_step / step_ms,
                              "frac": gf_step / step_ms / PEAK_FP32_MFMA_TFLOPS,
                              "note": "whole forward pass incl. every non-GEMM kernel: algorithmic FLOPs (SURVEY 8d: 16.8 GF/frame) / "
                                      "measured time / fp32-MFMA peak"},
                     "mfma_ms_per_step": sum(ms) / prof_steps, "mfma_time_frac_of_step": sum(ms) / prof_steps / step_ms},
        "kernels": kernels,
    }
    if world == 1 and not args.no_cpu_baseline:
        torch.set_num_threads(usable_cores())
        sd = procedural_state_dict(shapes_of(model.state_dict()))
        cb = synthetic_batch(args.batch, "cpu", seed=420)
        cperson, ccloth = torch.cat([cb["agnostic"], cb["densepose"]], 1), cb["cloth"]
        uhp = dict(n_frames_total=1, self_attn=True, num_attn=2, activation="gelu", flow_warp=False)
        times = []
        with torch.no_grad():
            for _ in range(2):
                oracle.unet_mask_forward(sd, cperson, ccloth, uhp)
            t_begin = time.perf_counter()
            while len(times) < max(args.cpu_iters, 5) and (len(times) < 3 or time.perf_counter() - t_begin < 45.0):
                t0 = time.perf_counter()
                oracle.unet_mask_forward(sd, cperson, ccloth, uhp)
                times.append(time.perf_counter() - t0)
        times.sort()
        med = times[len(times) // 2]
        out["cpu_baseline"] = {"value": args.batch / med, "unit": "frames/s", "cores": torch.get_num_threads(), "cpu": cpu_model(),
                               "kind": "port",
                               "sample": f"2 warm-up + {len(times)} timed UnetMaskModel forward passes of the oracle (bs={args.batch}, "
                                         f"256x192) with PyTorch CPU fp32, median {med * 1e3:.0f} ms (min {times[0] * 1e3:.0f}, "
                                         f"max {times[-1] * 1e3:.0f})"}
    print(json.dumps(flatten_roofline(out)), flush=True)


def hparams(**kw):
    base = dict(n_frames_total=1, cloth_inputs=["cloth"], is_train=True, ngf=64, grid_size=5, fine_height=256,
                fine_width=192, self_attn=True, num_attn=2, flow_warp=False, activation="gelu", display_count=10 ** 9,
                pen_flow_mask=1.0, lr=1e-4, keep_epochs=5, decay_epochs=5, allow_random_vgg=True)
    base.update(kw)
    return argparse.Namespace(**base)


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def flatten_roofline(out):
    """Scalar copies of the nested roofline entries (the driver's BENCH parser keeps scalars only): the step-level fraction,
    the MFMA aggregate, and one `hbm_<kernel>_gbs` / `_frac` pair per bandwidth-bound kernel of the `hbm` table."""
    import re

    r = out["roofline"]
    step = r.get("step") or {}
    r["step_gflop"] = step.get("algorithmic_gflop_per_step")
    r["step_tflops"] = step.get("achieved")
    r["step_frac"] = step.get("frac")
    for name, row in ((r.get("hbm") or {}).get("kernels") or {}).items():
        short = re.sub(r"[^a-z0-9]+", "_", name.split("(")[0].strip().lower()).strip("_")
        r[f"hbm_{short}_gbs"] = row["GB/s"]
        r[f"hbm_{short}_frac"] = row["frac_of_peak"]
    for name, row in (out.get("kernels") or {}).items():
        out[f"kernel_{name}_tflops"] = round(row["tflops"], 2)
        out[f"kernel_{name}_ms_per_step"] = round(row["total_ms_per_step"], 4)
    return out


def usable_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup quota, capped at 64
    (one socket's worth; PyTorch's CPU convolutions stop scaling well before that)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(config, batch_size, iters=5, warmup=2, budget_s=45.0):
    """The oracle (CPU restatement of the reference path) on this host's cores: the same step as the GPU leg,
    2 warm-up + >= 5 timed steps, median (SURVEY 8d / BASELINE.md §3).  Bounded: the timed loop stops early once
    `budget_s` is spent (never below 3 timed steps)."""
    from oracle import shineon_oracle as oracle
    from oracle.procedural import procedural_state_dict, shapes_of

    torch.set_num_threads(usable_cores())
    log(f"cpu_baseline: {torch.get_num_threads()} threads on {cpu_model()}")
    nfr = 5 if config == "c5" else 1
    batch = synthetic_batch(batch_size, "cpu", n_frames=nfr)
    if nfr > 1:
        batch = {k: (v.reshape(v.shape[0], -1, *v.shape[3:]) if isinstance(v, torch.Tensor) and v.dim() == 5 else v)
                 for k, v in batch.items()}
    steps = []
    if config in ("c2", "c4"):
        warp_sd = procedural_state_dict(shapes_of(WarpModel(hparams(person_inputs=["agnostic", "cocopose"])).state_dict()))
        wp = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in warp_sd.items()}
        optw = torch.optim.Adam([v for v in wp.values() if v.requires_grad], 1e-4)
        consts = oracle.tps_constants(256, 192, 5)
        whp = dict(person_inputs=["agnostic", "cocopose"], cloth_inputs=["cloth"])
    if config in ("c3", "c4", "c5"):
        extra = dict(n_frames_total=5, flow_warp=True) if config == "c5" else {}
        unet_sd = procedural_state_dict(shapes_of(UnetMaskModel(hparams(person_inputs=["agnostic", "densepose"], **extra)).state_dict()))
        up = {k: v.clone().requires_grad_(k.startswith("unet.")) for k, v in unet_sd.items()}
        optu = torch.optim.Adam([v for v in up.values() if v.requires_grad], 1e-4)
        uhp = dict(n_frames_total=nfr, person_inputs=["agnostic", "densepose"], cloth_inputs=["cloth"], self_attn=True,
                   num_attn=2, activation="gelu", flow_warp=nfr > 1)

    def step():
        b2 = batch
        if config in ("c2", "c4"):
            optw.zero_grad()
            out = oracle.warp_losses(wp, batch, whp, consts)
            out["loss/G"].backward()
            optw.step()
            b2 = dict(batch)
            b2["cloth"] = out["warped_cloth"].detach()
        if config in ("c3", "c4", "c5"):
            optu.zero_grad()
            oracle.unet_mask_losses(up, b2, uhp)["loss/G"].backward()
            optu.step()

    t_begin = time.perf_counter()
    for _ in range(warmup):
        step()
    while len(steps) < iters or (len(steps) < 3):
        t0 = time.perf_counter()
        step()
        steps.append(time.perf_counter() - t0)
        if time.perf_counter() - t_begin > budget_s and len(steps) >= 3:
            break
    steps.sort()
    med = steps[len(steps) // 2]
    what = {"c4": "chained steps (WarpModel + UnetMaskModel fwd+bwd+Adam", "c2": "WarpModel steps (fwd+bwd+Adam",
            "c3": "UnetMaskModel steps (fwd+bwd+Adam", "c5": "5-frame flow_warp UnetMaskModel steps (fwd+bwd+Adam"}[config]
    return {"value": batch_size * nfr / med, "unit": "frames/s", "cores": torch.get_num_threads(), "cpu": cpu_model(), "kind": "port",
            "sample": f"{warmup} warm-up + {len(steps)} timed {what}, bs={batch_size}, 256x192) with PyTorch CPU fp32, "
                      f"median {med * 1e3:.0f} ms/step (min {steps[0] * 1e3:.0f}, max {steps[-1] * 1e3:.0f})"}


def hbm_table(dev, batch_size):
    """GB/s of the bandwidth-bound kernels at the step's own tensor sizes (stand-alone launches, 20 repetitions, HIP
    events on the launch stream), algorithmic bytes per DESIGN.md §3.2, against the 8 TB/s HBM peak."""
    from shineon_virtual_tryon_amd import ops

    L = pkg.lib()
    st = torch.cuda.current_stream().cuda_stream
    f = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    n = batch_size
    cases = {}
    npar = 41_704_826  # trainable parameters of the two models (19.06 M + 22.65 M)
    p, g, m, v = (f(npar) for _ in range(4))
    v.abs_()  # second moments are non-negative
    cases["adam_step (41.7 M params, 28 B/param)"] = (
        lambda: L.so_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), npar, 1e-4, 0.9, 0.999, 1e-8, 3, 1.0, st),
        npar * 28)
    x = f(n * 128 * 96, 128)
    y = torch.empty(n * 256 * 192, 128, device=dev)
    cases["upsample2x bilinear (128x96 -> 256x192, C=128)"] = (
        lambda: L.so_upsample2x_fwd(x.data_ptr(), 128, y.data_ptr(), 128, n, 128, 96, 128, st), (x.numel() + y.numel()) * 4)
    xn = ops.nhwc_empty(n, 128, 96, 64, dev)
    cases["instance norm fwd (128x96, C=64; 12 B/elem)"] = (lambda: ops.instance_norm(xn), xn.numel() * 12)
    xv = f(2 * n * 256 * 192, 64)
    yv = torch.empty(2 * n * 128 * 96, 64, device=dev)
    cases["maxpool2x2 (VGG relu1_2, 2B images, C=64)"] = (
        lambda: L.so_maxpool2_fwd(xv.data_ptr(), 64, yv.data_ptr(), 64, 2 * n, 256, 192, 64, st), (xv.numel() + yv.numel()) * 4)
    src = f(n, 22, 256, 192)
    dst = torch.empty(n * 256 * 192, 24, device=dev)
    cases["nchw -> nhwc (person input, 22 -> 24 ch)"] = (
        lambda: L.so_nchw_to_nhwc(src.data_ptr(), dst.data_ptr(), 24, n, 22, 24, 256 * 192, st), (src.numel() + dst.numel()) * 4)
    a_ = f(n * 256 * 192, 64)
    b_ = torch.empty_like(a_)
    cases["gelu fwd (256x192 rows, C=64; 8 B/elem)"] = (
        lambda: L.so_act_fwd(a_.data_ptr(), 64, b_.data_ptr(), 64, a_.shape[0], 64, 3, 0.0, st), a_.numel() * 8)
    out = {}
    for name, (fn, nbytes) in cases.items():
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        gbs = nbytes / (us * 1e-6) / 1e9
        out[name] = {"bytes": nbytes, "avg_us": round(us, 2), "GB/s": round(gbs, 1), "frac_of_peak": round(gbs / PEAK_HBM_GBS, 3)}
    return out


def self_launch(n):
    """`--gpus n` given, no launcher environment: run `python -m torch.distributed.run --nproc-per-node n bench.py <same
    arguments>` as a child process, pass its stdout (rank 0's JSON line) and stderr through, return its exit code."""
    import socket
    import subprocess

    visible = torch.cuda.device_count()   # (counting devices does not initialise the GPU in this process)
    shared = os.environ.get("SHINEON_LOCAL_DEVICE") is not None   # functional runs: several ranks on one device over gloo
    if visible < n and not shared:
        log(f"bench.py: --gpus {n} but only {visible} GPU(s) are visible")
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: launching " + " ".join(cmd[1:]))
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="frames per GPU (BASELINE: 4)")
    ap.add_argument("--config", choices=("c4", "c1", "c2", "c3", "c5", "sams"), default="c4",
                    help="c4: chained warp->try-on step (headline); c1: UnetMaskModel forward only (BASELINE config 1, GPU + CPU leg); "
                         "c2: WarpModel alone; c3: UnetMaskModel alone; "
                         "c5: 5-frame flow_warp UnetMaskModel, bs=2 sequences (frames/s counts bs x 5 frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly (no hipGraph replay)")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--no-pipeline", action="store_true", help="c4: two sequential graphs (warp, then try-on) instead of "
                    "the two-stream schedule that overlaps the warp backward pass with the try-on stage")
    ap.add_argument("--no-hbm-table", action="store_true")
    ap.add_argument("--ramp-seconds", type=float, default=1.0, help="untimed: run the step for this long before the warm-up "
                    "steps so that the timed window starts at sustained clocks (0 = off)")
    ap.add_argument("--vgg-split-bf16", action="store_true", help="NON-HEADLINE experiment: the frozen VGG19 chain of the "
                    "perceptual loss on the bf16 matrix cores (fp32 = hi + mid bf16 planes, 3 MFMAs per product, fp32 "
                    "accumulate; csrc/sb16.hip); everything else stays exact fp32.  Reported with its own dtype string.")
    ap.add_argument("--plans", default="", help="extra igemm plans file: loaded on top of the committed one if present, "
                    "and every plan known at the end is written back to it")
    args = ap.parse_args()

    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus > 1 and env_world == 0:
        # `python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) and relay rank 0's line.  Decided
        # here, before this process has made any GPU call - the children are new processes, nothing is exec'ed over a
        # process that has touched the device.
        sys.exit(self_launch(args.gpus))
    if env_world and env_world != args.gpus:
        log(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks")
        sys.exit(2)

    trainer = Trainer(graph=not args.no_graph, overlap=True)   # joins the process group under torchrun
    rank, world = trainer.rank, trainer.world
    if world != args.gpus:   # what the process group itself reports must be what was asked for
        log(f"bench.py: --gpus {args.gpus} but the process group has {world} ranks")
        sys.exit(2)
    coll = so_trainer._collective()   # > 1 rank, or the one-rank RCCL group of SHINEON_SINGLE_RANK_GROUP=1 (functional run)
    dev = trainer.device
    L = pkg.lib()
    from shineon_virtual_tryon_amd import _lib as so_lib

    if so_lib.PLANS_LOADED:
        log(f"{so_lib.PLANS_LOADED[1]} committed igemm plans loaded from {os.path.relpath(so_lib.PLANS_LOADED[0], ROOT)}")
    if args.plans and os.path.exists(args.plans):
        log(f"loaded {L.so_igemm_plans_load(args.plans.encode())} igemm plans from {args.plans}")

    if args.vgg_split_bf16:
        from shineon_virtual_tryon_amd import ops as so_ops

        so_ops.VGG_SPLIT_BF16 = True
    torch.manual_seed(420)
    cfg = args.config
    if cfg == "sams":
        # --batch 4 (the default) is also the reference's default for this model (sams_model.py:40): ~50 GiB of HBM
        if "--steps" not in sys.argv:
            args.steps, args.warmup = 3, 1
        run_sams(args, trainer, L)
        if coll:
            dist.barrier()
            dist.destroy_process_group()
        return
    if cfg == "c1":
        run_c1(args, trainer, L)
        if coll:
            dist.barrier()
            dist.destroy_process_group()
        return
    if cfg == "c5" and args.batch == 4:
        args.batch = 2   # BASELINE config 5: bs = 2 per GPU
    nfr = 5 if cfg == "c5" else 1
    batch = synthetic_batch(args.batch, dev, seed=420, start=rank * args.batch, n_frames=nfr)
    warp = unet = None
    if cfg in ("c2", "c4"):
        warp = WarpModel(hparams(person_inputs=["agnostic", "cocopose"])).to(dev).train()
        warp.global_step = 1
    if cfg in ("c3", "c4", "c5"):
        extra = dict(n_frames_total=5, flow_warp=True) if cfg == "c5" else {}
        unet = UnetMaskModel(hparams(person_inputs=["agnostic", "densepose"], **extra)).to(dev).train()
        unet.global_step = 1

    if cfg == "c4":
        schedule = "eager" if args.no_graph else ("sequential" if args.no_pipeline else "auto")
        engine = trainer.build_chained_step(warp, unet, batch, schedule=schedule, log=log)
        step, flush, launch = (lambda: engine()), engine.flush, engine.launch_description
        eager_step = lambda: engine.eager_step()  # noqa: E731
        join = engine.synchronize
    else:
        model = warp if cfg == "c2" else unet
        (opt,), _ = model.configure_optimizers()
        broadcast_parameters(model, optimizer=opt)
        engine = TrainStep(model, opt, batch, graph=not args.no_graph, overlap=True)
        step, flush = (lambda: engine(batch)), engine.flush
        launch = "eager" if args.no_graph else "one hipGraph (forward+backward); Adam and all-reduce eager"
        if engine.exchange is not None:
            launch += "; " + engine.exchange.describe()
        eager_step = lambda: (engine.flush(), engine._eager(batch), engine.flush())  # noqa: E731
        join = lambda: (engine.flush(), torch.cuda.synchronize())  # noqa: E731
    log(f"{L.so_igemm_plan_count()} igemm plans in use")
    if coll:
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else "-"
        except Exception:  # noqa: BLE001
            rccl = "?"
        how = (engine.exchange.describe() if cfg != "c4" and engine.exchange is not None else
               "warp: " + engine.exw.describe() + "; try-on: " + engine.exu.describe() if cfg == "c4" and engine.exu is not None else
               "4 buckets each, started after the graph")
        log(f"rank {rank}/{world}: backend {dist.get_backend()} (RCCL {rccl}), NCCL_ALGO={os.environ.get('NCCL_ALGO', 'default')}, "
            f"NCCL_PROTO={os.environ.get('NCCL_PROTO', 'default')} (set NCCL_DEBUG=INFO for RCCL's own ring / tree report), "
            f"gradient slabs " + ", ".join(f"{o.flat_grads.numel() * 4 / 1e6:.1f} MB" for o in
                                           ([engine.optw, engine.optu] if cfg == "c4" else [opt])) + f": {how}")

    n_ramp = ramp_clocks(step, args.ramp_seconds)
    flush()
    log(f"rank {rank}/{world}: models built, clock ramp {n_ramp} untimed steps in {args.ramp_seconds:.1f} s, warm-up {args.warmup} steps")
    for i in range(args.warmup):
        t_w = time.perf_counter()
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i}: {1e3 * (time.perf_counter() - t_w):.1f} ms")
    flush()  # the timed region starts with no update pending

    def fence():
        torch.cuda.synchronize()
        if coll:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    if args.no_graph:
        L.so_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    flush()  # the last Adam update belongs to the timed steps
    fence()
    elapsed = time.perf_counter() - t0
    L.so_prof_enable(0)
    log(f"timed {args.steps} steps: {1e3 * elapsed / args.steps:.2f} ms/step")
    prof_steps = args.steps
    if not args.no_graph:
        # Kernels inside a replayed graph cannot be bracketed individually, so the per-kernel HIP-event timing
        # (roofline figure) is taken on the SAME kernels launched eagerly right after the timed region.
        prof_steps = min(args.steps, 5)
        join()
        eager_step()
        fence()
        L.so_prof_enable(1)
        for _ in range(prof_steps):
            eager_step()
        fence()
        L.so_prof_enable(0)
    exposed_ms = None
    if coll:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # What the gradient exchange costs on this node: the SAME steps once more with the collectives switched off (after
        # the timed region; the ranks' weights diverge from here on, only the clock is read).  exposed = with - without.
        reducers = ([x for x in (engine.redw, engine.redu, engine.exw, engine.exu) if x is not None] if cfg == "c4" else
                    [engine.exchange if engine.exchange is not None else engine.reducer])
        for r_ in reducers:
            r_.active = False
        for _ in range(2):
            step()
        flush()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        flush()
        fence()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exposed_ms = max(0.0, 1e3 * (elapsed - float(t.item())) / args.steps)
        log(f"gradient exchange exposed: {exposed_ms:.3f} ms/step ({1e3 * elapsed / args.steps:.3f} with, "
            f"{1e3 * float(t.item()) / args.steps:.3f} without collectives)")

    ms = (ctypes.c_float * NKEYS)()
    fl = (ctypes.c_float * NKEYS)()
    cnt = (ctypes.c_int * NKEYS)()
    by = (ctypes.c_double * NKEYS)()
    L.so_prof_collect_bytes(ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(cnt), ctypes.addressof(by))

    if rank == 0:
        kernels = {KEY_NAMES[k]: {"launches": cnt[k], "avg_us": 1e3 * ms[k] / cnt[k], "total_ms_per_step": ms[k] / prof_steps,
                                  "tflops": fl[k] / (ms[k] * 1e-3) / 1e12,
                                  "algorithmic_bytes_per_launch": (by[k] / cnt[k]) if by[k] > 0 else None}
                   for k in range(NKEYS) if cnt[k] > 0}
        dom0 = max(range(NKEYS), key=lambda k: ms[k])
        traffic = None  # HBM bytes per launch of the dominant instantiation, from the committed PMC passes
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            sect = json.load(open(tpath)).get(cfg, {})   # (the PMC passes see kernel symbols: Winograd-domain GEMMs = gemm_*)
            # bytes per launch scale with the batch: the section only applies to the batch its PMC passes ran at
            # (`_batch`; sections written before round 5 were taken at the configuration's default batch)
            if sect.get("_batch", 2 if cfg == "c5" else 4) == args.batch:
                traffic = (sect.get(KEY_NAMES[dom0]) or sect.get(KEY_NAMES[dom0].replace("winograd_gemm_", "gemm_"), {})).get("hbm_bytes_per_launch")
        dom, roof = dominant_roofline(
            ms, fl, by, cnt, traffic,
            "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, bytes per launch)",
            ("hip events, eager launches in the timed region" if args.no_graph else
             f"hip events on the same kernels launched eagerly for {prof_steps} steps right after the graph-replayed timed region"))
        for k_ in kernels.values():
            k_["executed_tflops"] = k_["tflops"]
        if "winograd_fused" in kernels:
            kernels["winograd_fused"]["executed_tflops"] = kernels["winograd_fused"]["tflops"] / WINOGRAD_FACTOR
        mfma_ms = sum(ms) / prof_steps
        step_ms = 1e3 * elapsed / args.steps
        gf_step = GF_PER_FRAME[cfg] * args.batch if GF_PER_FRAME[cfg] else sum(fl) / prof_steps / 1e9
        step_tflops = gf_step / step_ms  # GF / ms = TFLOP/s (per GPU: weak scaling)
        out = {
            "metric": "try-on frames/sec (fwd+bwd) at 256x192 bs=4" if cfg != "c5" else
                      "try-on frames/sec (fwd+bwd) at 256x192, n_frames=5 video batch bs=2",
            "value": world * args.batch * nfr * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "n_ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("fp32" if not args.vgg_split_bf16 else
                      "fp32; frozen VGG19 chain as 3x bf16-split MFMA (hi*hi + hi*mid + mid*hi, fp32 accumulate) - non-headline"),
            "data": "synthetic",
            "config": {"workload": WORKLOADS[cfg], "config": cfg, "launch": launch, "batch_per_gpu": args.batch,
                       "global_batch": world * args.batch, "frames_per_sample": nfr, "parallelism": f"dp{world}",
                       "step_api": "shineon_virtual_tryon_amd.trainer." + ("ChainedTrainStep" if cfg == "c4" else "TrainStep"),
                       "exchange_exposed_ms": exposed_ms,
                       "exchange": (None if not coll else
                                    (engine.exchange.describe() if cfg != "c4" and getattr(engine, "exchange", None) is not None
                                     else "warp: " + engine.exw.describe() + "; try-on: " + engine.exu.describe()
                                     if cfg == "c4" and getattr(engine, "exu", None) is not None
                                     else "flat slab in 4 buckets after the graph, hidden behind the other model's graph")),
                       # True: the warp model's BatchNorm running statistics are broadcast in synchronize(), not per step
                       # (DDP broadcasts per forward; rounds 1-3 timed that broadcast inside the c4 step)
                       "lazy_buffers": getattr(engine, "lazy_buffers", None) if cfg == "c4" else None,
                       "untimed_clock_ramp_s": args.ramp_seconds,
                       "pipeline_gain_ms": getattr(engine, "pipeline_gain_ms", None) if cfg == "c4" else None,
                       "exchange_probe_ms": getattr(engine, "exchange_ms", None) if cfg == "c4" else None},
            "roofline": {
                **roof,
                "step": {"algorithmic_gflop_per_step": gf_step, "achieved": step_tflops,
                         "frac": step_tflops / PEAK_FP32_MFMA_TFLOPS,
                         "note": "whole step incl. every non-GEMM kernel, Adam and launch gaps: algorithmic FLOPs (SURVEY 8d) / "
                                 "measured step time / fp32-MFMA peak"},
                "mfma_ms_per_step": mfma_ms, "mfma_time_frac_of_step": mfma_ms / step_ms,
                # what the matrix pipes really execute per step (every MFMA launch's own 2MNK, the fused Winograd kernel at
                # 1 / 2.25 of its direct-convolution figure) / step time / peak: the pipe utilisation of the WHOLE step
                "step_executed_gflop": (sum(fl) - fl[WINOGRAD_KEY] * (1 - 1 / WINOGRAD_FACTOR)) / prof_steps / 1e9,
                "step_executed_frac": (sum(fl) - fl[WINOGRAD_KEY] * (1 - 1 / WINOGRAD_FACTOR)) / prof_steps / 1e9 / step_ms
                / PEAK_FP32_MFMA_TFLOPS,
                "all_mfma_tflops": sum(fl) / (sum(ms) * 1e-3) / 1e12 if sum(ms) > 0 else 0.0,
                "all_mfma_executed_tflops": (sum(fl) - fl[WINOGRAD_KEY] * (1 - 1 / WINOGRAD_FACTOR)) / (sum(ms) * 1e-3) / 1e12
                if sum(ms) > 0 else 0.0,
            },
            "kernels": kernels,
        }
        if not args.no_hbm_table:
            out["roofline"]["hbm"] = {"peak_GB/s": PEAK_HBM_GBS, "kernels": hbm_table(dev, args.batch)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.batch, args.cpu_iters)
        print(json.dumps(flatten_roofline(out)), flush=True)
    if args.plans and rank == 0:
        log(f"saved {L.so_igemm_plans_save(args.plans.encode())} igemm plans to {args.plans}")
    if coll:
        torch.cuda.synchronize()     # nothing of the engines' streams in flight when the communicators go away
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
